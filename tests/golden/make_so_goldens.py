#!/usr/bin/env python3
"""Transcribe the reference's pinned successive-orders radiances into a JSON fixture.

Source of the vectors (read as data, never executed):
  /root/reference/tests/engine/test_1d_solver_regression.py:140-197
      the `("successive_orders", 3, False, np.array([...]))` parameter of test_1d_solver_radiance_regression: exact single
      scatter + SuccessiveOrdersLegacy, NSTOKES = 3, 3 iterations, 26 incoming / 26 outgoing directions, 2 wavelengths x 4 rays
      x (I, Q, U); compared there with rtol 5e-7 / atol 2e-13
  the inputs are those of `_setup_1d` (:9-109), rebuilt by tests/scenarios.py:regression_so (geometry, rays and optical
  properties as regression_1d; a2_2 = 3, b1_2 = -sqrt(6)/2; num_streams 8, num_singlescatter_moments 16, num_sza 2)

Run (in the authoring container, where /root/reference exists):
    python tests/golden/make_so_goldens.py
"""
import json
import re
from pathlib import Path

SRC = Path("/root/reference/tests/engine/test_1d_solver_regression.py")
OUT = Path(__file__).with_name("so_regression.json")


def main():
    text = SRC.read_text()
    start = text.index('"successive_orders",\n            3,')
    end = text.index("def test_1d_solver_radiance_regression")
    block = text[start:end]
    numbers = [float(x) for x in re.findall(r"-?\d+\.\d+(?:[eE][+-]?\d+)?", block)]
    assert len(numbers) == 2 * 4 * 3, len(numbers)
    radiance = [[numbers[(w * 4 + r) * 3:(w * 4 + r) * 3 + 3] for r in range(4)] for w in range(2)]
    OUT.write_text(json.dumps({
        "source": "tests/engine/test_1d_solver_regression.py:140-197 (successive_orders, NSTOKES = 3)",
        "rtol": 5e-7, "atol": 2e-13,
        "config": {"num_stokes": 3, "num_streams": 8, "num_singlescatter_moments": 16, "num_sza": 2,
                   "num_successive_orders_iterations": 3, "num_successive_orders_incoming": 26, "num_successive_orders_outgoing": 26,
                   "single_scatter_source": "exact", "multiple_scatter_source": "successive_orders_legacy"},
        "expected_radiance_wavel_los_stokes": radiance,
        "optical_depth_rows_of_the_do_case": [0, 2],
    }, indent=1) + "\n")
    print("wrote", OUT)


if __name__ == "__main__":
    main()
