import sys
import numpy as np
a, b = np.load(f"gpurun_out/park_{sys.argv[1]}.npz"), np.load(f"gpurun_out/park_{sys.argv[2]}.npz")
for k in a.files:
    print(k, "identical" if np.array_equal(a[k], b[k]) else f"DIFFERENT (digests {a[k][0]:.0f} / {b[k][0]:.0f}, max abs {a[k][1]:.3e} / {b[k][1]:.3e})")
