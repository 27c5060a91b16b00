#!/usr/bin/env python3
"""Transcribe the reference's polarised (NSTOKES = 3) DISCO known-answer tests into a JSON fixture.

Source of the vectors (read as data, never executed):
  /root/reference/cpp/lib/tests/sktran_disco/legacy/test_vector.cpp:31-277
      atmospheres, suns, lines of sight, expected Stokes vectors and tolerances of the four known-answer cases
      (scalar equivalence; Siewert's aerosol slab; Coulson's Rayleigh tables for albedo 0 and 0.8)
  /root/reference/cpp/lib/sktran_disco/sktran_do_testing.cpp:5-143
      how a layer spec becomes Greek coefficients: Henyey-Greenstein a1 = (2l+1) g^l; Rayleigh with depolarisation d
      (a1_0 = 1, a1_2 = (1-d)/(2+d), a2_2 = 6 (1-d)/(2+d), b1_2 = sqrt(6) (1-d)/(2+d)); Siewert's 12-term aerosol table
How a TestCase becomes engine input: cpp/lib/tests/sktran_disco/test_util.cpp:24-118 (leg_coeff order a1, a2, a3, -b1);
tests/scenarios.py:planeparallel_vector_case rebuilds those inputs from this fixture.

Run (in the authoring container, where /root/reference exists):
    python tests/golden/make_vector_goldens.py
"""
import json
import math
import re
from pathlib import Path

SRC = Path("/root/reference/cpp/lib/tests/sktran_disco/legacy/test_vector.cpp")
TESTING = Path("/root/reference/cpp/lib/sktran_disco/sktran_do_testing.cpp")
OUT = Path(__file__).with_name("do_vector_planeparallel.json")

TOL = {"SKDO_FPC_EPS": 1e-6, "SKDO_FPC_COULSEN_EPS": 1e-5, "1.2 * SKDO_FPC_COULSEN_EPS": 1.2e-5}


def num(expr: str) -> float:
    expr = expr.strip().replace("PI", repr(math.pi))
    assert re.fullmatch(r"[0-9eE+\-*/. ()]+", expr), expr
    return float(eval(expr))  # arithmetic on literals only


def strip(text):
    return re.sub(r"//[^\n]*", "", text)


def parse_atmos(text):
    out = {}
    for m in re.finditer(r"TestAtmosphere<3>\s+(\w+)\s*=\s*std::vector<TestLayerSpec(\w+)>\(\s*\{(.*?)\}\s*\)\s*;", text, re.S):
        kind = m.group(2)
        layers = [[num(x) for x in l.split(",")] for l in re.findall(r"TestLayerSpec\w+\(\{([^}]*)\}\)", strip(m.group(3)))]
        out[m.group(1)] = {"kind": kind, "layers": layers}
    return out


def parse_suns(text):
    out = {}
    for m in re.finditer(r"TestSolarSpec\s+(\w+)\s*=\s*\{(.*?)\}\s*;", text, re.S):
        body = strip(m.group(2)).replace("{", "").replace("}", "")
        vals = [num(x) for x in body.split(",") if x.strip()]
        out[m.group(1)] = {"csz": vals[0], "saz": vals[1], "direct": vals[2]}
    return out


def parse_los(text):
    out = {}
    for m in re.finditer(r"std::vector<TestLOS>\s+(\w+)\s*=\s*\{(.*?)\}\s*;", text, re.S):
        pairs = re.findall(r"\{([^{}]*?),([^{}]*?)\}", strip(m.group(2)))
        out[m.group(1)] = [[num(a), num(b)] for a, b in pairs]
    return out


def siewert_table():
    """The 12 Greek coefficients of Siewert's aerosol problem as the NSTOKES = 3 harness assigns them."""
    text = TESTING.read_text()
    body = text[text.index("TestLayer<3>::assignSiewertLayer"):text.index("TestLayer<4>::assignSiewertLayer")]
    tab = {}
    for name in ("a1", "a2", "a3", "b1"):
        vals = re.findall(r"temp\[(\d+)\]\." + name + r"\s*=\s*(-?[0-9.]+);", body)
        assert [int(i) for i, _ in vals] == list(range(12)), name
        tab[name] = [float(v) for _, v in vals]
    return tab


def main():
    src = SRC.read_text()
    head, *blocks = re.split(r"\nTEST_CASE\(", src)
    g_atm, g_sun, g_los = parse_atmos(head), parse_suns(head), parse_los(head)
    cases = []
    for b in blocks:
        name = re.match(r'\s*"([^"]*)"', b).group(1)
        tc = re.search(r"TestCase<3>\s+testcase\(\s*(\d+)\s*,\s*(\w+)\s*,\s*(\w+)\s*,\s*([0-9.]+)\s*,\s*(\w+)\s*,\s*(\w+)\s*\)", b)
        rad = re.search(r"std::vector<double>\s+correct_radiances\s*=\s*\{(.*?)\}\s*;", b, re.S)
        if not tc or not rad:
            continue  # the 1-vs-3-layer self-consistency case carries no stored vector
        atm = {**g_atm, **parse_atmos(b)}[tc.group(3)]
        sun = {**g_sun, **parse_suns(b)}[tc.group(2)]
        los = {**g_los, **parse_los(b)}[tc.group(5)]
        expected = [num(x) for x in strip(rad.group(1)).split(",") if x.strip()]
        if tc.group(6) == "correct_stokes":  # scalar radiances promoted to (I, 0, 0)
            expected = [v for x in expected for v in (x, 0.0, 0.0)]
        assert len(expected) == 3 * len(los), name
        req = re.search(r"REQUIRE\(diff < ([^)]*)\)", b)
        cases.append({"name": name, "nstr": int(tc.group(1)), "sun": sun, "albedo": float(tc.group(4)),
                      "layer_kind": atm["kind"], "layers": atm["layers"], "los_coszen_az": los,
                      "expected_stokes": expected, "tolerance": TOL[req.group(1).strip()]})
    OUT.write_text(json.dumps({"source": "cpp/lib/tests/sktran_disco/legacy/test_vector.cpp",
                               "siewert_greek_coefficients": siewert_table(), "cases": cases}, indent=1))
    print(f"wrote {len(cases)} cases to {OUT}")
    for c in cases:
        print(" ", c["name"], c["nstr"], c["layer_kind"], len(c["los_coszen_az"]), c["tolerance"])


if __name__ == "__main__":
    main()
