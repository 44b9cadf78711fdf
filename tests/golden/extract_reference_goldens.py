#!/usr/bin/env python3
"""Extract the golden input/output VECTORS (data only, no code) that the reference's own unit
tests hold for the decomposition path, and write them to tests/golden/reference_goldens.json.

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
    dst = os.path.join(os.path.dirname(os.path.abspath(__file__)), "reference_goldens.json")
    with open(dst, "w") as f:
        json.dump(out, f, indent=1)
    print("wrote", dst, "with", len(out["cases"]), "cases")


if __name__ == "__main__":
    main()
