"""mmgen — MI355X-native chunk-generation path (hand-written HIP for gfx950 behind a C ABI).

Python here is plumbing only: device memory (torch), streams and torch.distributed.  The product is
``libmmgen.so`` (csrc/, C ABI in include/mmgen.h).  Importing this package never touches ``oracle/``.
"""
from .mmgen import MMGen, load_library, LIB_PATH, build  # noqa: F401
