#!/usr/bin/env python3
"""Extract the golden input/output VECTORS (data only, no code) that the reference's own unit
tests hold for the decomposition path, and write them to tests/golden/reference_goldens.json.

Source of the operator vectors ("operators"): <reference>/tests/src/test_TensorMassMatrix.cpp :21-262,
test_TensorRestriction.cpp :18-221, test_TensorProlongation.cpp :16-106 (dyadic sections).
Source of the vectors: <reference>/tests/src/test_decompose.cpp
  TEST_CASE("decomposition") :276-457  sections "1D/2D/4D, dyadic, uniform", "1D, dyadic, nonuniform"
  TEST_CASE("recomposition") :549-746  sections "1D/3D/4D, dyadic, uniform"

Each dyadic section holds one input vector `u_` and a list `expecteds`; entry L of `expecteds` is
the expected result for the hierarchy of shape (2^L+1)^N applied to the first (2^L+1)^N entries
of `u_`, in natural (unshuffled) node order (test_decompose.cpp:36-96).

Run in the build container only (the reference is not present on the GPU box):
    python tests/golden/extract_reference_goldens.py [/root/reference]
"""
import json
import os
import re
import sys


def _section(text, case, section):
    ci = text.index('TEST_CASE("%s"' % case)
    nxt = text.find("TEST_CASE(", ci + 1)
    body = text[ci: nxt if nxt > 0 else len(text)]
    si = body.index('SECTION("%s")' % section)
    nxt = body.find("SECTION(", si + 1)
    return body[si: nxt if nxt > 0 else len(body)]


def _braces(s, start):
    """Return the substring of the balanced {...} starting at s[start] == '{'."""
    depth = 0
    for i in range(start, len(s)):
        if s[i] == "{":
            depth += 1
        elif s[i] == "}":
            depth -= 1
            if depth == 0:
                return s[start: i + 1]
    raise ValueError("unbalanced braces")


def _parse_list(s):
    s = re.sub(r"//[^\n]*", "", s)                       # C++ line comments
    s = s.replace("{", "[").replace("}", "]")
    s = re.sub(r",\s*\]", "]", s)                         # trailing commas
    s = re.sub(r"(?<![\d.])(-?)\.(\d)", r"\g<1>0.\2", s)   # .5 -> 0.5
    s = re.sub(r"(\d)\.(?!\d)", r"\1.0", s)              # 1. -> 1.0
    return json.loads(s)


def _vector(sec, name):
    m = re.search(r"\b%s\s*=\s*\{" % re.escape(name), sec)
    return _parse_list(_braces(sec, m.end() - 1))


def _dtype(sec, name):
    m = re.search(r"std::vector<(?:std::vector<)?(float|double)>+\s*%s\b" % re.escape(name), sec)
    return m.group(1)


def _num_list(text):
    """A C++ brace initializer whose entries are arithmetic on literals (`2.6 / 6 + 6.0 / 6`, `-3`,
    `1. / 18`) as nested lists of floats: the entries are evaluated, nothing else is."""
    text = re.sub(r"//[^\n]*", "", text)

    def parse(i):
        assert text[i] == "{"
        i += 1
        items, cur = [], ""
        while True:
            c = text[i]
            if c == "{":
                sub, i = parse(i)
                items.append(sub)
                cur = ""
            elif c in ",}":
                if cur.strip():
                    expr = cur.strip()
                    assert re.fullmatch(r"[0-9eE.+\-*/ \n\t()]+", expr), expr
                    items.append(float(eval(expr, {"__builtins__": {}})))
                cur = ""
                i += 1
                if c == "}":
                    return items, i
            else:
                cur += c
                i += 1

    val, _ = parse(0)
    while isinstance(val, list) and len(val) == 1 and isinstance(val[0], list):
        val = val[0]  # std::array's double braces
    return val


def _init(sec, name, nth=0):
    """The nth initializer `name = {...}` / `name{...}` in sec."""
    ms = list(re.finditer(r"\b%s\b\s*=?\s*\{" % re.escape(name), sec))
    m = ms[nth]
    return _num_list(_braces(sec, m.end() - 1))


def _between(text, a, b=None, start=0):
    i = text.index(a, start)
    j = text.index(b, i + len(a)) if b else len(text)
    return text[i:j]


def _operators(ref):
    """Known answers of the MGARD-CPU operators on DYADIC grids (where MGARD-CPU and MGARD-X define the
    same hierarchy), default and custom spacing. Each entry: the constituent operator applied along
    `dimension` on the level-`l` nodes of the lines starting at `multiindices`, everything in natural
    node order. Transformations the test code applies to its literals (a scaling by 1/48, a division
    of three entries by 12) are applied here and cited."""
    src = os.path.join(ref, "tests", "src")
    ops = []
    # ---- tests/src/test_TensorMassMatrix.cpp ----
    t = open(os.path.join(src, "test_TensorMassMatrix.cpp")).read()
    case = _between(t, 'TEST_CASE("constituent mass matrices"', 'TEST_CASE("tensor product mass matrices"')
    sec = _between(case, 'SECTION("1D and default spacing")', 'SECTION("1D and nondyadic")')
    u = _init(sec, "u_")
    e3 = [x / 48 for x in _init(sec, "expected", 0)]          # :38-39 blas::scal(ndof, 1/48, expected)
    e1 = _init(sec, "expected", 1)
    for i in range(3):                                         # :54-56 expected.at(4 * i) /= 12
        e1[4 * i] /= 12
    ops.append({"op": "mass", "source": "test_TensorMassMatrix.cpp:21-62", "dtype": "float", "shape": [9],
                "coords": None, "u": u, "l": 3, "dimension": 0, "multiindices": [[0]], "expected": e3})
    ops.append({"op": "mass", "source": "test_TensorMassMatrix.cpp:21-62", "dtype": "float", "shape": [9],
                "coords": None, "u": u, "l": 1, "dimension": 0, "multiindices": [[0]], "expected": e1})
    sec = _between(case, 'SECTION("2D and custom spacing")')
    coords = _num_list(_braces(sec, sec.index("{5, 5}, {{{") + len("{5, 5}, ")))
    u = _init(sec, "u_")
    ops.append({"op": "mass", "source": "test_TensorMassMatrix.cpp:96-150", "dtype": "double", "shape": [5, 5],
                "coords": coords, "u": u, "l": 2, "dimension": 0, "multiindices": [[0, 0], [0, 3]],
                "expected": _init(sec, "expected", 0)})
    ops.append({"op": "mass", "source": "test_TensorMassMatrix.cpp:151-194", "dtype": "double", "shape": [5, 5],
                "coords": coords, "u": u, "l": 2, "dimension": 1, "multiindices": [[1, 0], [2, 0]],
                "expected": _init(sec, "expected", 1)})
    last = _init(sec, "expected", 2)                           # :203-211 only the last row changes
    ops.append({"op": "mass", "source": "test_TensorMassMatrix.cpp:195-213", "dtype": "double", "shape": [5, 5],
                "coords": coords, "u": u, "l": 1, "dimension": 1, "multiindices": [[4, 0]],
                "expected": u[:20] + last})
    case = _between(t, 'TEST_CASE("tensor product mass matrices"', "namespace {")
    coords = _num_list(_braces(case, case.index("{3, 3}, {{{") + len("{3, 3}, ")))
    u = _init(case, "u_")
    for k, l in ((0, 1), (1, 0)):
        ops.append({"op": "tensor_mass", "source": "test_TensorMassMatrix.cpp:217-262", "dtype": "double",
                    "shape": [3, 3], "coords": coords, "u": u, "l": l, "expected": _init(case, "expected", k)})
    # ---- tests/src/test_TensorRestriction.cpp ----
    t = open(os.path.join(src, "test_TensorRestriction.cpp")).read()
    case = _between(t, 'TEST_CASE("constituent restrictions"', "namespace {")
    sec = _between(case, 'SECTION("1D and default spacing")', 'SECTION("1D and custom spacing and nondyadic")')
    u, ex = _init(sec, "u_"), _init(sec, "expecteds")
    for i, l in enumerate((3, 2, 1)):                          # :33-34 l = 3, 2, 1 <-> expecteds[0, 1, 2]
        ops.append({"op": "restriction", "source": "test_TensorRestriction.cpp:19-47", "dtype": "float", "shape": [9],
                    "coords": None, "u": u, "l": l, "dimension": 0, "multiindices": [[0]], "expected": ex[i]})
    sec = _between(case, 'SECTION("2D and custom spacing")')
    coords = _num_list(_braces(sec, sec.index("{3, 3}, {{{") + len("{3, 3}, ")))
    u = _init(sec, "u_")
    ex0, ex1 = _init(sec, "expecteds", 0), _init(sec, "expecteds", 1)
    for mi, e in zip(([0, 0], [0, 1]), ex0):
        ops.append({"op": "restriction", "source": "test_TensorRestriction.cpp:78-113", "dtype": "double",
                    "shape": [3, 3], "coords": coords, "u": u, "l": 1, "dimension": 0, "multiindices": [mi],
                    "expected": e})
    for mi, e in zip(([1, 0], [2, 0]), ex1):
        ops.append({"op": "restriction", "source": "test_TensorRestriction.cpp:114-133", "dtype": "double",
                    "shape": [3, 3], "coords": coords, "u": u, "l": 1, "dimension": 1, "multiindices": [mi],
                    "expected": e})
    case = _between(t, 'TEST_CASE("tensor product restrictions"', "std::default_random_engine generator(445624)")
    coords = _num_list(_braces(case, case.index("{3, 3}, {{{") + len("{3, 3}, ")))
    ops.append({"op": "tensor_restriction", "source": "test_TensorRestriction.cpp:200-221", "dtype": "double",
                "shape": [3, 3], "coords": coords, "u": _init(case, "u_"), "l": 1,
                "expected": _init(case, "expected")})
    # ---- tests/src/test_TensorProlongation.cpp ----
    t = open(os.path.join(src, "test_TensorProlongation.cpp")).read()
    case = _between(t, 'TEST_CASE("constituent prolongations"', "namespace {")
    sec = _between(case, 'SECTION("1D and default spacing")', 'SECTION("2D and custom spacing")')
    u, ex = _init(sec, "u_"), _init(sec, "expecteds")
    for i, l in enumerate((3, 2, 1)):
        ops.append({"op": "prolongation_addition", "source": "test_TensorProlongation.cpp:17-48", "dtype": "float",
                    "shape": [9], "coords": None, "u": u, "l": l, "dimension": 0, "multiindices": [[0]],
                    "expected": ex[i]})
    sec = _between(case, 'SECTION("2D and custom spacing")', 'SECTION("2D and nondyadic")')
    coords = _num_list(_braces(sec, sec.index("{3, 3}, {{{") + len("{3, 3}, ")))
    u = _init(sec, "u_")
    ex0, ex1 = _init(sec, "expecteds", 0), _init(sec, "expecteds", 1)
    for mi, e in zip(([0, 0], [0, 1]), ex0):
        ops.append({"op": "prolongation_addition", "source": "test_TensorProlongation.cpp:50-83", "dtype": "double",
                    "shape": [3, 3], "coords": coords, "u": u, "l": 1, "dimension": 0, "multiindices": [mi],
                    "expected": e})
    for mi, e in zip(([1, 0], [2, 0]), ex1):
        ops.append({"op": "prolongation_addition", "source": "test_TensorProlongation.cpp:84-106", "dtype": "double",
                    "shape": [3, 3], "coords": coords, "u": u, "l": 1, "dimension": 1, "multiindices": [mi],
                    "expected": e})
    return {"source": "tests/src/test_Tensor{MassMatrix,Restriction,Prolongation}.cpp (CODARcode/MGARD v1.6.0)",
            "note": "dyadic grids only (on those MGARD-CPU's hierarchy is MGARD-X's); natural node order",
            "cases": ops}


def main():
    ref = sys.argv[1] if len(sys.argv) > 1 else "/root/reference"
    path = os.path.join(ref, "tests", "src", "test_decompose.cpp")
    text = open(path).read()
    out = {"source": "tests/src/test_decompose.cpp (CODARcode/MGARD v1.6.0)", "cases": []}
    for case, sections in (("decomposition", [("1D, dyadic, uniform", 1), ("2D, dyadic, uniform", 2),
                                              ("4D, dyadic, uniform", 4)]),
                           ("recomposition", [("1D, dyadic, uniform", 1), ("3D, dyadic, uniform", 3),
                                              ("4D, dyadic, uniform", 4)])):
        for name, ndim in sections:
            sec = _section(text, case, name)
            out["cases"].append({
                "kind": case, "name": name, "ndim": ndim, "dtype": _dtype(sec, "u_"),
                "u": _vector(sec, "u_"), "expecteds": _vector(sec, "expecteds"),
            })
    sec = _section(text, "decomposition", "1D, dyadic, nonuniform")
    coords = _parse_list(_braces(sec, sec.index("coordinates = {{") + len("coordinates = {")))
    u = _parse_list(_braces(sec, sec.index("u_ = {{") + len("u_ = {")))
    expected = _parse_list(_braces(sec, sec.index("expected = {{") + len("expected = {")))
    out["cases"].append({"kind": "decomposition", "name": "1D, dyadic, nonuniform", "ndim": 1,
                         "dtype": "float", "coordinates": [coords], "u": u, "expected": expected})
    # known answers of the scalar quantizer (tests/src/test_LinearQuantizer.cpp:94-121):
    # quantize = round(x / quantum) to nearest, dequantize = quantum * n
    qtext = open(os.path.join(ref, "tests", "src", "test_LinearQuantizer.cpp")).read()
    qsec = qtext[qtext.index('TEST_CASE("quantization of a range"'):]
    s1 = qsec[qsec.index('SECTION("basic quantization iteration")'):qsec.index('SECTION("basic dequantization iteration")')]
    s2 = qsec[qsec.index('SECTION("basic dequantization iteration")'):qsec.index('SECTION("quantization inverts dequantization")')]
    quantum1 = float(re.search(r"quantizer\(([0-9.]+)\)", s1).group(1))
    xs1 = _parse_list(_braces(s1, s1.index("xs = {") + len("xs = ")))
    ns1 = _parse_list(_braces(s1, s1.index("std::vector<int>({") + len("std::vector<int>(")))
    quantum2 = float(re.search(r"dequantizer\(([0-9.]+)\)", s2).group(1))
    ns2 = _parse_list(_braces(s2, s2.index("ns = {") + len("ns = ")))
    xs2 = _parse_list(_braces(s2, s2.index("std::vector<float>({") + len("std::vector<float>(")))
    out["quantizer"] = {
        "source": "tests/src/test_LinearQuantizer.cpp:94-111 (CODARcode/MGARD v1.6.0)",
        "quantize": {"quantum": quantum1, "dtype": "double", "x": xs1, "n": ns1},
        "dequantize": {"quantum": quantum2, "dtype": "float", "n": ns2, "x": xs2},
    }
    out["operators"] = _operators(ref)
    dst = os.path.join(os.path.dirname(os.path.abspath(__file__)), "reference_goldens.json")
    with open(dst, "w") as f:
        json.dump(out, f, indent=1)
    print("wrote", dst, "with", len(out["cases"]), "cases")


if __name__ == "__main__":
    main()
