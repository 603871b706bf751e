#!/usr/bin/env python3
"""Numeric literals of the reference's arithmetic functions, per function and per switch case, as DATA (numbers only):
tests/golden/ref_literals.json = {"<file>::<function>[::<case>]": [sorted distinct float32 values]}.

The function bodies of biomeFuncs.hpp / featurePlacement.hpp / rng.hpp / chunk.cu cannot be compiled here (cuda / thrust headers), and the
oracle and the device code restate them from the same reading.  A mistyped or dropped constant is the kind of common-mode error no
HIP-vs-oracle test can see; this fixture lets a CPU test hold every section of the oracle (same function and case names) and of the device
code (MMBIO_ / MMF_ / MMCF_ / MMCB_ case labels) to the set of constants the reference's own text uses: every value the reference writes
in a section must appear in ours.  (Control flow and operation order stay unpinned: DESIGN.md §2.)

A second fixture, tests/golden/ref_skeletons.json, holds ONE SHA-256 per section: the digest of the section's token stream after a
fixed normalisation (`skeleton`: comments, qualifiers, braces, (float) casts and printf statements dropped, numbers by value, a table of
renamed helpers).  The oracle's same-named section is normalised the same way; equal digests mean the oracle's section is the reference's
statement for statement - control flow, operation order, operands - up to that table.  A digest is not the text: nothing of the source
can be read back from it.

Run in the build container: python tools/extract_ref_literals.py   (reads /root/reference, writes both fixtures).
The section parser (`sections`, `literals`) is imported by tests/test_ref_literals.py to parse OUR sources the same way.
"""
import json
import os
import re
import sys

import numpy as np

REF = "/root/reference/src"
NUM = re.compile(r"(?<![\w.])(?:0[xX][0-9a-fA-F]*\.?[0-9a-fA-F]*(?:[pP][-+]?\d+)?|\d+\.\d*(?:[eE][-+]?\d+)?|\.\d+(?:[eE][-+]?\d+)?|\d+(?:[eE][-+]?\d+)?)(?:f|F|u|U|ull|ULL|ll|LL|l|L)?(?![\w.])")


def strip_comments(text):
    text = re.sub(r"/\*.*?\*/", " ", text, flags=re.S)
    return resolve_conditionals(re.sub(r"//[^\n]*", " ", text))


def resolve_conditionals(text):
    """keep the active branch of #if / #ifdef / #ifndef ... #else ... #endif for the object-like macros the (comment-free) text itself
    defines (`#define DEBUG_USE_CONTRIBUTION_FILL_METHOD 0` at the top of chunk.cu; a commented-out #define is not defined); the
    directive lines themselves are dropped.  Conditions: NAME, !NAME, defined(NAME); anything else is kept as written."""
    defs = {}
    for m in re.finditer(r"^[ \t]*#[ \t]*define[ \t]+(\w+)[ \t]+(-?\d+)[ \t]*$", text, flags=re.M):
        defs[m.group(1)] = int(m.group(2))
    out, stack = [], []                                     # stack of (active_before, this_branch_taken, currently_active)
    for line in text.split("\n"):
        m = re.match(r"[ \t]*#[ \t]*(ifdef|ifndef|if|else|endif)\b[ \t]*(.*)", line)
        active = all(s[2] for s in stack)
        if not m:
            if active:
                out.append(line)
            continue
        kind, cond = m.group(1), m.group(2).strip()
        if kind in ("ifdef", "ifndef"):
            val = (cond in defs) != (kind == "ifndef")
            stack.append([active, val, val])
        elif kind == "if":
            neg = cond.startswith("!")
            name = cond.lstrip("! ").strip()
            dm = re.match(r"defined\s*\(?\s*(\w+)\s*\)?$", name)
            if dm:
                val = dm.group(1) in defs
            elif re.fullmatch(r"\w+", name):
                val = bool(defs.get(name, 0))
            else:
                out.append(line)                                # not ours to decide: leave the construct in the token stream
                stack.append([active, True, True])
                continue
            val = val != neg
            stack.append([active, val, val])
        elif kind == "else" and stack:
            stack[-1][2] = not stack[-1][1]
        elif kind == "endif" and stack:
            stack.pop()
    return "\n".join(out)


def literals(code):
    """sorted distinct numeric literal values of a piece of code, as float32 (hex ints as ints); template arguments count like any number"""
    vals = set()
    for m in NUM.finditer(code):
        tok = m.group(0)
        if tok.lower().startswith("0x"):
            body = tok.rstrip("uUlL")
            if "p" in body.lower():
                v = float.fromhex(body.rstrip("fF"))               # hex float (0x1.8p+1f)
            else:
                v = float(int(body, 16))                           # hex integer: a trailing f is a digit here
        else:
            v = float(tok.rstrip("fFuUlL"))
        vals.add(float(np.float32(v)))
    return sorted(vals)


def body_after(text, start):
    """text of the brace block that starts at the first '{' after `start`"""
    i = text.index("{", start)
    depth = 0
    for j in range(i, len(text)):
        if text[j] == "{":
            depth += 1
        elif text[j] == "}":
            depth -= 1
            if depth == 0:
                return text[i:j + 1]
    raise ValueError("unbalanced braces")


STATEMENT = "statement"


def sections(text, signature, case_prefixes=(), anchors=None):
    """{section: code}.  `signature` is a regex matching the start of the function definition; with case_prefixes the body is cut at
    `case <prefix>NAME:` labels (consecutive labels share the code that follows; code before the first label is section '').
    anchors = (start regex, end regex or None): only the part of the body from the first match of `start` up to the first match of `end`
    behind it (or the body's end) counts - the arithmetic core of a CUDA kernel without its thread-index preamble and barriers."""
    out = {}
    if case_prefixes == STATEMENT:                            # a file-scope definition that is one statement: the matches themselves
        code = " ".join(m.group(0) for m in re.finditer(signature, text))
        return {"": code} if code else {}
    for m in re.finditer(signature, text):
        try:
            body = body_after(text, m.end() - 1 if text[m.end() - 1] == "{" else m.end())
        except ValueError:
            continue
        if anchors:
            a = re.search(anchors[0], body)
            if not a:
                continue
            e = re.search(anchors[1], body[a.start():]) if anchors[1] else None
            body = body[a.start():a.start() + e.start()] if e else body[a.start():]
        if not case_prefixes:
            out[""] = out.get("", "") + body
            continue
        label = re.compile(r"case\s+(?:" + "|".join(re.escape(p) for p in case_prefixes) + r")(\w+)\s*:")
        pos, names = 0, [""]
        pieces = []
        for lm in label.finditer(body):
            between = body[pos:lm.start()]
            if between.strip():
                pieces.append((names, between))
                names = []
            names = names + [lm.group(1)]
            pos = lm.end()
        pieces.append((names, body[pos:]))
        for nm, code in pieces:
            for n in nm:
                out[n] = out.get(n, "") + code
    return out


# ---- statement skeletons -------------------------------------------------------------------------------------------------------
TOKEN = re.compile(r"0[xX][0-9a-fA-F]+[uUlL]*|\d+\.\d*(?:[eE][-+]?\d+)?[fF]?|\.\d+(?:[eE][-+]?\d+)?[fF]?|\d+(?:[eE][-+]?\d+)?[fFuUlL]*|[A-Za-z_]\w*|::|->|<<=|>>=|<<|>>"
                   r"|<=|>=|==|!=|&&|\|\||\+=|-=|\*=|/=|%=|\+\+|--|[-+*/%<>=!&|^~?:;,.()\[\]{}]")
# helpers the oracle names differently (its own deterministic libm, glm look-alikes with a g_ prefix, typed overloads)
RENAME = {"g_smoothstep": "smoothstep", "g_mix": "mix", "g_clamp": "clamp", "g_max": "max", "g_min": "min", "g_floor": "floor", "g_fract": "fract",
          "g_length": "length", "g_normalize": "normalize", "g_dot": "dot", "g_cross": "cross", "g_distance": "distance", "g_mod": "mod",
          "g_abs": "abs", "g_sqrt": "sqrt", "mm_powf": "powf", "mm_sinf": "sin", "mm_cosf": "cos", "mm_acosf": "acos", "mm_atan2f": "atan2",
          "mm_fmodf": "fmod", "mm_sqrtf": "sqrt", "mm_sincosf": "sincosf", "sinf": "sin", "cosf": "cos", "acosf": "acos", "atan2f": "atan2",
          "fmodf": "fmod", "fabsf": "abs", "fmaxf": "max", "fminf": "min", "fmax": "max", "fmin": "min", "floorf": "floor", "sqrtf": "sqrt",
          "isInRangeF": "isInRange", "isInRangeI": "isInRange", "g_ceil": "ceil", "ceilf": "ceil", "g_angle": "angle", "g_sin": "sin",
          "hash_u32": "hash", "Rng": "default_random_engine", "g_radians": "radians",
          # C++ leaves the order of evaluation of call arguments unspecified; where the reference draws from one random stream in several
          # arguments of one vec3(...) the oracle fixes the canonical left-to-right order with a braced list behind this macro (mmo_vec.h)
          "vec3_ltr": "vec3"}
DROPPED = {"const", "__device__", "__host__", "static", "inline", "glm", "thrust", "std", "::"}
CONTROL = {"if", "for", "while", "switch"}


def canonical_blocks(toks):
    """Block structure made canonical instead of dropped: the body of every if / else / for / while / do / switch is wrapped in ONE pair of
    braces whether the source wrote them or not (`if (c) a;` == `if (c) { a; }`, `else if` == `else { if ... }`), and braces that are not
    the body of a control statement (a scope around a case's statements, the function's own, the stray closers of a section cut out of
    a switch) are dropped.  `if (c) { a; b; }` and `if (c) a; b;` therefore differ: which statements a condition or a loop governs is
    part of the digest.  Braces inside an expression statement (initialiser lists) stay as written."""
    n = len(toks)
    out = []

    def close_paren(i):                                       # toks[i] == "(" -> index after its ")"
        depth = 0
        while i < n:
            depth += toks[i] == "("
            depth -= toks[i] == ")"
            i += 1
            if depth == 0:
                break
        return i

    def body(i):
        out.append("{")
        if i < n and toks[i] == "{":
            i += 1
            while i < n and toks[i] != "}":
                i = statement(i)
            i += 1
        else:
            i = statement(i)
        out.append("}")
        return i

    def statement(i):
        if i >= n:
            return i
        t = toks[i]
        if t == "{":                                          # a plain scope
            i += 1
            while i < n and toks[i] != "}":
                i = statement(i)
            return i + 1
        if t in CONTROL:
            out.append(t)
            i += 1
            if i < n and toks[i] == "(":
                j = close_paren(i)
                out.extend(toks[i:j])
                i = j
            i = body(i)
            if t == "if" and i < n and toks[i] == "else":
                out.append("else")
                i = body(i + 1)
            return i
        if t == "else":                                       # (a section that starts between an if and its else)
            out.append("else")
            return body(i + 1)
        if t == "do":
            out.append("do")
            i = body(i + 1)
            while i < n and toks[i] != ";":
                out.append(toks[i])
                i += 1
            out.append(";")
            return i + 1
        if t == "case" or (t == "default" and i + 1 < n and toks[i + 1] == ":"):
            while i < n and toks[i] != ":":
                out.append(toks[i])
                i += 1
            out.append(":")
            return i + 1
        depth = 0                                             # expression / declaration statement: up to its ';'
        while i < n:
            t = toks[i]
            if t in "([{" and len(t) == 1:
                depth += 1
            elif t in ")]}" and len(t) == 1:
                if depth == 0:
                    return i                                  # the closer of an enclosing block: not ours
                depth -= 1
            out.append(t)
            i += 1
            if t == ";" and depth == 0:
                break
        return i

    i = 0
    while i < n:
        if toks[i] == "}":                                    # closer of a block that was opened before this section
            i += 1
            continue
        i = statement(i)
    return out


def _number(tok):
    if tok.lower().startswith("0x"):
        return "#%d" % int(tok.rstrip("uUlL"), 16)
    body = tok.rstrip("fFuUlL")
    if any(c in body for c in ".eE") or tok[-1] in "fF":
        return "#%r" % float(np.float32(float(body)))
    return "#%d" % int(body)                                         # an integer literal stays an integer: 2 / 3 is not 2.f / 3.f


def skeleton(code):
    """normalised token list of a piece of (comment-free) code"""
    code = re.sub(r"^([ \t]*#[ \t]*(?:define|undef)\b[^\n]*)$", r"\1 ;", code, flags=re.M)      # a function-local macro is a statement of its own
    toks = [m.group(0) for m in TOKEN.finditer(code)]
    out, i = [], 0
    while i < len(toks):
        t = toks[i]
        if t == "(" and i + 2 < len(toks) and toks[i + 1] == "float" and toks[i + 2] == ")":      # (float) casts the oracle makes explicit
            i += 3
            continue
        if t not in DROPPED:
            out.append(_number(t) if (t[0].isdigit() or (t[0] == "." and len(t) > 1)) else RENAME.get(t, t))
        i += 1
    txt = " " + " ".join(out) + " "
    txt = re.sub(r" printf \( [^;]* \) ;", " ", txt)                    # "reached an unreachable section" diagnostics
    txt = txt.replace(" default : break ;", " ").replace(" pragma unroll ", " ").replace(" __builtin_unreachable ( ) ;", " ")
    # the two statements the canonical semantics add to member functions of the reference (oracle/mmo_stages.cpp)
    txt = txt.replace(" CANONICAL_RETURN_FALSE ;", " ").replace(" CANONICAL_DECORATOR_RANGE ( pos ) ;", " ")
    txt = txt.replace(" CANONICAL_NO_LAYER_FOUND ( thisLayerIdx , blockPtr )", " ").replace(" CANONICAL_CAVE_LAYER_OVERFLOW ( storeIdx )", " ")
    txt = re.sub(r" \( void \) \w+ ;", " ", txt)
    txt = txt.replace(" . r ", " . x ").replace(" . g ", " . y ").replace(" . b ", " . z ")      # glm colour aliases of the components
    return canonical_blocks(txt.split())


def skeleton_digest(code):
    import hashlib
    toks = skeleton(code)
    return {"sha256": hashlib.sha256(" ".join(toks).encode()).hexdigest(), "tokens": len(toks)}


# (file, key, signature regex, case prefixes)
REFERENCE_SECTIONS = [
    ("terrain/biomeFuncs.hpp", "getSingleBiomeNoise", r"float\s+getSingleBiomeNoise\s*\([^)]*\)\s*\{", ()),
    ("terrain/biomeFuncs.hpp", "getBiomeNoise", r"BiomeNoise\s+getBiomeNoise\s*\([^)]*\)\s*\{", ()),
    ("terrain/biomeFuncs.hpp", "getSingleCaveBiomeNoise", r"float\s+getSingleCaveBiomeNoise\s*\([^)]*\)\s*\{", ()),
    ("terrain/biomeFuncs.hpp", "getCaveBiomeNoise", r"CaveBiomeNoise\s+getCaveBiomeNoise\s*\([^)]*\)\s*\{", ()),
    ("terrain/biomeFuncs.hpp", "getHeight", r"float\s+getHeight\s*\([^)]*\)\s*\{", ("Biome::",)),
    ("terrain/biomeFuncs.hpp", "biomeBlockPreProcess", r"bool\s+biomeBlockPreProcess\s*\([^)]*\)\s*\{", ("Biome::",)),
    ("terrain/biomeFuncs.hpp", "biomeBlockPostProcess", r"bool\s+biomeBlockPostProcess\s*\([^)]*\)\s*\{", ("Biome::",)),
    ("terrain/biomeFuncs.hpp", "caveBiomeBlockPostProcess", r"bool\s+caveBiomeBlockPostProcess\s*\([^)]*\)\s*\{", ("CaveBiome::",)),
    ("terrain/featurePlacement.hpp", "sdCappedCylinder", r"float\s+sdCappedCylinder\s*\([^)]*\)\s*\{", ()),
    ("terrain/featurePlacement.hpp", "isInRasterizedLine", r"bool\s+isInRasterizedLine\s*\([^)]*\)\s*\{", ()),
    ("terrain/featurePlacement.hpp", "jungleLeaves", r"bool\s+jungleLeaves\s*\([^)]*\)\s*\{", ()),
    ("terrain/featurePlacement.hpp", "getCrystalRadius", r"float\s+getCrystalRadius\s*\([^)]*\)\s*\{", ()),
    ("terrain/featurePlacement.hpp", "isInCrystal", r"bool\s+isInCrystal\s*\([^)]*\)\s*\{", ()),
    ("terrain/featurePlacement.hpp", "getRandomCrystalBlock", r"Block\s+getRandomCrystalBlock\s*\([^)]*\)\s*\{", ()),
    ("terrain/featurePlacement.hpp", "placeFeature", r"bool\s+placeFeature\s*\([^)]*\)\s*\{", ("Feature::",)),
    ("terrain/featurePlacement.hpp", "placeCaveFeature", r"bool\s+placeCaveFeature\s*\([^)]*\)\s*\{", ("CaveFeature::",)),
    ("terrain/biomeFuncs.hpp", "getRandomBiome", r"Biome\s+getRandomBiome\s*\([^)]*\)\s*\{", ()),
    ("terrain/biomeFuncs.hpp", "applySingleBiomeNoise", r"void\s+applySingleBiomeNoise\s*\([^)]*\)\s*\{", ()),
    ("terrain/biomeFuncs.hpp", "getBiomeWeight", r"float\s+getBiomeWeight\s*\([^)]*\)\s*\{", ()),
    ("terrain/biomeFuncs.hpp", "getCaveBiomeWeight", r"float\s+getCaveBiomeWeight\s*\([^)]*\)\s*\{", ()),
    ("terrain/biomeFuncs.hpp", "getCaveBiome", r"CaveBiome\s+getCaveBiome\s*\([^)]*\)\s*\{", ()),
    ("terrain/featurePlacement.hpp", "sdSphere", r"float\s+sdSphere\s*\([^)]*\)\s*\{", ()),
    ("terrain/featurePlacement.hpp", "opSubtraction", r"float\s+opSubtraction\s*\([^)]*\)\s*\{", ()),
    ("terrain/featurePlacement.hpp", "opOnion", r"float\s+opOnion\s*\([^)]*\)\s*\{", ()),
    ("terrain/featurePlacement.hpp", "deCasteljau", r"void\s+deCasteljau\s*\([^)]*\)\s*\{", ()),
    ("util/rng.hpp", "manhattanLength", r"int\s+manhattanLength\s*\([^)]*\)\s*\{", ()),
    ("util/rng.hpp", "manhattanDistance", r"int\s+manhattanDistance\s*\([^)]*\)\s*\{", ()),
    ("util/rng.hpp", "isInRange", r"bool\s+isInRange\s*\([^)]*\)\s*\{", ()),
    ("util/rng.hpp", "isPosInRange", r"bool\s+isPosInRange\s*\([^)]*\)\s*\{", ()),
    ("util/rng.hpp", "getRatio", r"float\s+getRatio\s*\([^)]*\)\s*\{", ()),
    ("util/rng.hpp", "saturate", r"float\s+saturate\s*\([^)]*\)\s*\{", ()),
    ("util/rng.hpp", "isSaturated", r"(?:float|bool)\s+isSaturated\s*\([^)]*\)\s*\{", ()),
    ("util/rng.hpp", "calculateLineParams", r"bool\s+calculateLineParams\s*\([^)]*\)\s*\{", ()),
    ("util/rng.hpp", "fbm", r"float\s+fbm\s*\([^)]*\)\s*\{", ()),
    ("util/rng.hpp", "worley2", r"float\s+worley\s*\(\s*vec2[^)]*\)\s*\{", ()),
    ("util/rng.hpp", "worley3", r"float\s+worley\s*\(\s*vec3[^)]*\)\s*\{", ()),
    ("util/rng.hpp", "hash", r"unsigned\s+int\s+hash\s*\([^)]*\)\s*\{", ()),
    ("util/rng.hpp", "makeSeededRandomEngine", r"makeSeededRandomEngine\s*\([^)]*\)\s*\{", ()),
    ("util/rng.hpp", "rand1From1", r"float\s+rand1From1\s*\([^)]*\)\s*\{", ()),
    ("util/rng.hpp", "rand1From2", r"float\s+rand1From2\s*\([^)]*\)\s*\{", ()),
    ("util/rng.hpp", "rand1From3", r"float\s+rand1From3\s*\([^)]*\)\s*\{", ()),
    ("util/rng.hpp", "rand2From2", r"vec2\s+rand2From2\s*\([^)]*\)\s*\{", ()),
    ("util/rng.hpp", "rand2From3", r"vec2\s+rand2From3\s*\([^)]*\)\s*\{", ()),
    ("util/rng.hpp", "rand3From2", r"vec3\s+rand3From2\s*\([^)]*\)\s*\{", ()),
    ("util/rng.hpp", "rand3From3", r"vec3\s+rand3From3\s*\([^)]*\)\s*\{", ()),
    ("util/rng.hpp", "fbm2From2", r"vec2\s+fbm2From2\s*\([^)]*\)\s*\{", ()),
    ("util/rng.hpp", "fbm3From3", r"vec3\s+fbm3From3\s*\([^)]*\)\s*\{", ()),
    ("util/rng.hpp", "simplex2From2", r"vec2\s+simplex2From2\s*\([^)]*\)\s*\{", ()),
    ("util/rng.hpp", "specialCaveNoise", r"float\s+specialCaveNoise\s*\([^)]*\)\s*\{", ()),
    ("terrain/chunk.cu", "shouldGenerateCaveAtBlock", r"bool\s+shouldGenerateCaveAtBlock\s*\([^)]*\)\s*\{", ()),
    ("terrain/chunk.cu", "getStratifiedMaterialThickness", r"float\s+getStratifiedMaterialThickness\s*\([^)]*\)\s*\{", ()),
    ("terrain/chunk.cu", "isFeaturePos", r"bool\s+isFeaturePos\s*\([^)]*\)\s*\{", ()),
    ("terrain/chunk.cu", "tryGenerateCaveFeaturePlacement", r"bool\s+Chunk::tryGenerateCaveFeaturePlacement\s*\([^)]*\)\s*\{", ()),
    ("terrain/chunk.cu", "tryPlaceSingleDecorator", r"void\s+Chunk::tryPlaceSingleDecorator\s*\([^)]*\)\s*\{", ()),
    ("terrain/chunk.cu", "generateColumnFeaturePlacements", r"void\s+Chunk::generateColumnFeaturePlacements\s*\([^)]*\)\s*\{", ()),
    ("terrain/chunk.cu", "placeDecorators", r"void\s+Chunk::placeDecorators\s*\([^)]*\)\s*\{", ()),
    ("terrain/chunk.cu", "chunkFillPlaceBlock", r"void\s+chunkFillPlaceBlock\s*\([^{]*\)\s*\{", ()),
    # kernGenerateCaves, phase by phase (its four barriers and the warp shuffle cut it into seven pieces; the oracle states each as a
    # function of its own and runs every thread of the block through one before the next starts)
    ("terrain/chunk.cu", "kernGenerateCaves.init", r"void\s+kernGenerateCaves\s*\([^{]*\)\s*\{", ()),
    ("terrain/chunk.cu", "kernGenerateCaves.weights", r"void\s+kernGenerateCaves\s*\([^{]*\)\s*\{", ()),
    ("terrain/chunk.cu", "kernGenerateCaves.filled", r"void\s+kernGenerateCaves\s*\([^{]*\)\s*\{", ()),
    ("terrain/chunk.cu", "kernGenerateCaves.flips", r"void\s+kernGenerateCaves\s*\([^{]*\)\s*\{", ()),
    ("terrain/chunk.cu", "kernGenerateCaves.compact", r"void\s+kernGenerateCaves\s*\([^{]*\)\s*\{", ()),
    ("terrain/chunk.cu", "kernGenerateCaves.store", r"void\s+kernGenerateCaves\s*\([^{]*\)\s*\{", ()),
    ("terrain/chunk.cu", "kernGenerateCaves.biomes", r"void\s+kernGenerateCaves\s*\([^{]*\)\s*\{", ()),
    # the arithmetic that is written inline in the CUDA kernels and host stages (no callee to pin): whole functions where they are plain
    # C++, the part between ANCHORS where the function is a kernel (its thread-index preamble, shared-memory staging copies and barriers
    # are orchestration; what a thread computes is not)
    ("terrain/chunk.cu", "kernGenerateHeightfield", r"void\s+kernGenerateHeightfield\s*\([^{]*\)\s*\{", ()),
    ("terrain/chunk.cu", "kernGenerateLayers", r"void\s+kernGenerateLayers\s*\([^{]*\)\s*\{", ()),
    ("terrain/chunk.cu", "kernDoErosion.stage", r"void\s+kernDoErosion\s*\([^{]*\)\s*\{", ()),
    ("terrain/chunk.cu", "kernDoErosion.relax", r"void\s+kernDoErosion\s*\([^{]*\)\s*\{", ()),
    ("terrain/chunk.cu", "copyLayers", r"void\s+copyLayers\s*\([^{]*\)\s*\{", ()),
    ("terrain/chunk.cu", "fixBackwardStratifiedLayers", r"void\s+Chunk::fixBackwardStratifiedLayers\s*\(\s*\)\s*\{", ()),
    ("terrain/chunk.cu", "kernFill", r"void\s+kernFill\s*\([^{]*\)\s*\{", ()),
    ("terrain/chunk.cu", "heightBoundsMinMax", r"void\s+heightBoundsMinMax\s*\([^{]*\)\s*\{", ()),
    ("terrain/chunk.cu", "Chunk.fill.lists", r"void\s+Chunk::fill\s*\([^{]*\)\s*\{", ()),
    # SURVEY 8f-2, the mesh build that follows the path: Chunk::createVBOs (plain host C++, whole function) and the static tables in front
    # of it (chunk.cu:1753-1776; file-scope definitions: a brace initialiser is a body like any other, a one-statement definition is
    # taken as the statement itself)
    ("terrain/chunk.cu", "createVBOs", r"void\s+Chunk::createVBOs\s*\(\s*\)\s*\{", ()),
    ("terrain/chunk.cu", "createVBOs.xShapedPosOffset", r"float\s+xShapedPosOffset\s*=[^;]*;", STATEMENT),
    ("terrain/chunk.cu", "createVBOs.xShapedVertPositions", r"xShapedVertPositions\s*=\s*\{", ()),
    ("terrain/chunk.cu", "createVBOs.xShapedFaceNormals", r"vec3\s+xShapedFaceNormal[12]\s*=[^;]*;", STATEMENT),
    ("terrain/chunk.cu", "createVBOs.directionVertPositions", r"directionVertPositions\s*=\s*\{", ()),
    ("terrain/chunk.cu", "createVBOs.uvOffsets", r"\buvOffsets\s*=\s*\{", ()),
]
# key -> (start, end): the section is the part of the function body from `start` up to `end` (None = the body's end)
ANCHORS = {
    "kernGenerateCaves.init": (r"const\s+int\s+globalX\s*=", r"__syncthreads"),                                # chunk.cu:824-844
    "kernGenerateCaves.weights": (r"if\s*\(\s*y\s*<\s*numOceanAndBeachBiomes", r"__syncthreads"),               # chunk.cu:846-850
    "kernGenerateCaves.filled": (r"int\s+isThisFilled\s*=", r"__syncthreads"),                                  # chunk.cu:854-855
    "kernGenerateCaves.flips": (r"int\s+isNextFilled\s*=", r"__syncthreads"),                                   # chunk.cu:859-860
    "kernGenerateCaves.compact": (r"const\s+int\s+startLoadIdx\s*=", r"int\s+startStoreIdx"),                   # chunk.cu:873-886
    "kernGenerateCaves.store": (r"int\s+startStoreIdx\s*=", r"if\s*\(\s*y\s*<\s*MAX_CAVE_LAYERS_PER_COLUMN\s*\)"),    # chunk.cu:887-908
    "kernGenerateCaves.biomes": (r"CaveLayer\s*&\s*caveLayer\s*=", None),                                      # chunk.cu:913-936
    "kernGenerateHeightfield": (r"const\s+int\s+idx\s*=", None),                                             # chunk.cu:160-184
    "kernGenerateLayers": (r"float\s+totalMaterialWeights\s*\[", None),                                      # chunk.cu:346-414
    "kernDoErosion.stage": (r"const\s+int\s+localX\s*=\s*threadIdx", r"__syncthreads"),                       # chunk.cu:487-555
    "kernDoErosion.relax": (r"float\s+newLayerStart\s*=\s*thisLayerStart", r"__syncthreads"),                 # chunk.cu:558-590
    "kernFill": (r"const\s+float\s+height\s*=\s*shared_layersAndHeight", None),                              # chunk.cu:1427-1509
    "Chunk.fill.lists": (r"ivec2\s+allFeaturesHeightBounds\s*=\s*ivec2", r"const\s+dim3\s+blockSize3d"),      # chunk.cu:1555-1601
}


# where the oracle's section ends differently from the reference's (the oracle hands a value on where the reference's function goes on)
ORACLE_ANCHORS = {
    "kernGenerateCaves.compact": (r"const\s+int\s+startLoadIdx\s*=", r"warp_numFlips\s*\["),
    "kernGenerateCaves.store": (r"int\s+startStoreIdx\s*=", None),
}


def oracle_anchors(key):
    return ORACLE_ANCHORS.get(key, ANCHORS.get(key))


# where the oracle restates each reference file
ORACLE_FILES = {"biomeFuncs.hpp": ["oracle/mmo_biome.h", "oracle/mmo_biome.cpp"], "featurePlacement.hpp": ["oracle/mmo_features.cpp"],
                "rng.hpp": ["oracle/mmo_noise.h"], "chunk.cu": ["oracle/mmo_stages.cpp", "oracle/mmo_mesh.cpp"]}

# the oracle's spelling of a reference signature (free functions instead of Chunk:: members, its own kernel-less stage names)
ORACLE_SIGNATURES = {"rng.hpp::hash": r"uint32_t\s+hash_u32\s*\([^)]*\)\s*\{",
                     # a kernel with a barrier is two functions in the oracle (every thread runs the first, then every thread the second)
                     "chunk.cu::kernDoErosion.stage": r"void\s+kernDoErosion_stage\s*\([^{]*\)\s*\{",
                     "chunk.cu::kernDoErosion.relax": r"void\s+kernDoErosion_relax\s*\([^{]*\)\s*\{",
                     "chunk.cu::Chunk.fill.lists": r"void\s+Chunk_fill\s*\([^{]*\)\s*\{",
                     **{"chunk.cu::kernGenerateCaves." + ph: r"void\s+kernGenerateCaves_" + ph + r"\s*\([^{]*\)\s*\{"
                        for ph in ("init", "weights", "filled", "flips", "compact", "store", "biomes")}}


def oracle_signature(key, sig):
    return ORACLE_SIGNATURES.get(key, sig)


def main(out_path):
    out = {}
    cache = {}
    for rel, key, sig, prefixes in REFERENCE_SECTIONS:
        if rel not in cache:
            cache[rel] = strip_comments(open(os.path.join(REF, rel)).read())
        secs = sections(cache[rel], sig, prefixes, ANCHORS.get(key))
        assert secs, (rel, key)
        for name, code in secs.items():
            vals = literals(code)
            if vals:
                out[f"{os.path.basename(rel)}::{key}" + (f"::{name}" if name else "")] = vals
    json.dump(out, open(out_path, "w"), indent=0, sort_keys=True)
    print(f"wrote {out_path}: {len(out)} sections, {sum(len(v) for v in out.values())} literal values")
    skel = {}
    for rel, key, sig, prefixes in REFERENCE_SECTIONS:
        for name, code in sections(cache[rel], sig, prefixes, ANCHORS.get(key)).items():
            skel[f"{os.path.basename(rel)}::{key}" + (f"::{name}" if name else "")] = skeleton_digest(code)
    skel_path = os.path.join(os.path.dirname(out_path), "ref_skeletons.json")
    json.dump(skel, open(skel_path, "w"), indent=0, sort_keys=True)
    print(f"wrote {skel_path}: {len(skel)} section digests")


if __name__ == "__main__":
    main(sys.argv[1] if len(sys.argv) > 1 else os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "ref_literals.json"))
