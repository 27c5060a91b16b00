#!/usr/bin/env python3
"""Transcribe the reference's scalar DISCO known-answer tests into a JSON fixture.

Source of the vectors (read as data, never executed):
  /root/reference/cpp/lib/tests/sktran_disco/legacy/test_scalar.cpp:34-800
Each TEST_CASE there defines a plane-parallel layered atmosphere (optical depth, SSA,
Henyey-Greenstein asymmetry per layer), a sun, a Lambertian albedo, a stream count,
35 (or 8) lines of sight and the 12-digit expected radiances.  How a TestCase becomes
engine input is described at cpp/lib/tests/sktran_disco/test_util.cpp:24-118 and
cpp/lib/sktran_disco/sktran_do_testing.cpp:5-14; tests/test_oracle_golden.py rebuilds
those inputs from this fixture.

Run (in the authoring container, where /root/reference exists):
    python tests/golden/make_scalar_goldens.py
"""
import json
import math
import re
from pathlib import Path

SRC = Path("/root/reference/cpp/lib/tests/sktran_disco/legacy/test_scalar.cpp")
OUT = Path(__file__).with_name("do_scalar_planeparallel.json")

TOL = {"SKDO_FPC_EPS": 1e-8, "SKDO_FPC_EPS_LOW_PRECISION": 1e-6}


def num(expr: str) -> float:
    expr = expr.strip().replace("PI", repr(math.pi))
    assert re.fullmatch(r"[0-9eE+\-*/. ()]+", expr), expr
    return float(eval(expr))  # arithmetic on literals only


def parse_atmos(text):
    out = {}
    for m in re.finditer(r"TestAtmosphere<1>\s+(\w+)\s*=\s*std::vector<TestLayerSpecHG>\(\s*\{(.*?)\}\s*\)\s*;", text, re.S):
        body = re.sub(r"//[^\n]*", "", m.group(2))
        layers = [[num(x) for x in l.split(",")] for l in re.findall(r"TestLayerSpecHG\(\{([^}]*)\}\)", body)]
        out[m.group(1)] = layers
    return out


def parse_suns(text):
    out = {}
    for m in re.finditer(r"TestSolarSpec\s+(\w+)\s*=\s*\{(.*?)\}\s*;", text, re.S):
        body = re.sub(r"//[^\n]*", "", m.group(2)).replace("{", "").replace("}", "")
        vals = [num(x) for x in body.split(",") if x.strip()]
        out[m.group(1)] = {"csz": vals[0], "saz": vals[1], "direct": vals[2]}
    return out


def parse_los(text):
    out = {}
    for m in re.finditer(r"std::vector<TestLOS>\s+(\w+)\s*=\s*\{(.*?)\}\s*;", text, re.S):
        pairs = re.findall(r"\{([^{}]*?),([^{}]*?)\}", m.group(2))
        out[m.group(1)] = [[num(a), num(b)] for a, b in pairs]
    return out


def main():
    src = SRC.read_text()
    head, *blocks = re.split(r"\nTEST_CASE\(", src)
    g_atm, g_sun, g_los = parse_atmos(head), parse_suns(head), parse_los(head)
    cases = []
    for b in blocks:
        name = re.match(r'\s*"([^"]*)"', b).group(1)
        tc = re.search(r"TestCase<1>\s+testcase\(\s*(\d+)\s*,\s*(\w+)\s*,\s*(\w+)\s*,\s*([0-9.]+)\s*,\s*(\w+)\s*,\s*correct_radiances\s*\)", b)
        rad = re.search(r"std::vector<double>\s+correct_radiances\s*=\s*\{(.*?)\}\s*;", b, re.S)
        if not tc or not rad:
            continue  # BRDF case (not implemented in the reference) and the 1-vs-3-layer self-consistency case
        atm = {**g_atm, **parse_atmos(b)}[tc.group(3)]
        sun = {**g_sun, **parse_suns(b)}[tc.group(2)]
        los = {**g_los, **parse_los(b)}[tc.group(5)]
        expected = [num(x) for x in rad.group(1).split(",") if x.strip()]
        assert len(expected) == len(los), name
        req = re.search(r"REQUIRE\(diff < (\w+)\)", b)
        cases.append({
            "name": name,
            "nstr": int(tc.group(1)),
            "sun": sun,
            "albedo": float(tc.group(4)),
            "layers_od_ssa_g": atm,
            "los_coszen_az": los,
            "expected": expected,
            "tolerance": TOL[req.group(1)] if req else None,
        })
    OUT.write_text(json.dumps({"source": "cpp/lib/tests/sktran_disco/legacy/test_scalar.cpp", "cases": cases}, indent=1))
    print(f"wrote {len(cases)} cases to {OUT}")


if __name__ == "__main__":
    main()
