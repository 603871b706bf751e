#!/usr/bin/env python3
"""Developer aid (build container only: reads /root/reference): where does the oracle's section stop being token-identical to the
reference's?  Prints, per section, the number of differing token runs and (with a section name) a short window of normalised tokens
around each difference.  Nothing it prints is stored; tests use the digests of tests/golden/ref_skeletons.json.

    python tools/skeleton_diff.py                 # one line per section
    python tools/skeleton_diff.py placeFeature::CORAL   # windows around each difference of the sections whose key contains the argument
"""
import difflib
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from extract_ref_literals import REF, REFERENCE_SECTIONS, sections, skeleton, strip_comments   # noqa: E402

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
from extract_ref_literals import ANCHORS, ORACLE_FILES, oracle_anchors, oracle_signature   # noqa: E402


def main():
    want = sys.argv[1] if len(sys.argv) > 1 else None
    cache_r, cache_o = {}, {}
    same = diff = 0
    for rel, key, sig, prefixes in REFERENCE_SECTIONS:
        base = os.path.basename(rel)
        if rel not in cache_r:
            cache_r[rel] = strip_comments(open(os.path.join(REF, rel)).read())
        if base not in cache_o:
            cache_o[base] = "\n".join(strip_comments(open(os.path.join(ROOT, p)).read()) for p in ORACLE_FILES[base])
        rs = sections(cache_r[rel], sig, prefixes, ANCHORS.get(key))
        os_ = sections(cache_o[base], oracle_signature(f"{base}::{key}", sig), prefixes, oracle_anchors(key))
        for name, code in rs.items():
            k = f"{base}::{key}" + (f"::{name}" if name else "")
            a, b = skeleton(code), skeleton(os_.get(name, ""))
            if a == b:
                same += 1
                if not want:
                    print(f"  same    {k}")
                continue
            diff += 1
            sm = difflib.SequenceMatcher(None, a, b, autojunk=False)
            ops = [o for o in sm.get_opcodes() if o[0] != "equal"]
            if not want:
                print(f"  DIFF {len(ops):3d} {k}   ({len(a)} / {len(b)} tokens)")
            elif want in k:
                print(f"== {k}: {len(ops)} differences ({len(a)} ref / {len(b)} oracle tokens)")
                for tag, i1, i2, j1, j2 in ops:
                    print("   ref: ", " ".join(a[max(0, i1 - 6):i1]), "[[", " ".join(a[i1:i2]), "]]", " ".join(a[i2:i2 + 4]))
                    print("   ours:", " ".join(b[max(0, j1 - 6):j1]), "[[", " ".join(b[j1:j2]), "]]", " ".join(b[j2:j2 + 4]))
                    print()
    print(f"{same} identical, {diff} differing")


if __name__ == "__main__":
    main()
