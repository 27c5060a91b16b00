import sys, traceback
sys.path.insert(0, "/root/repo")
import tests.test_gpu_parity as t
bad = []
for seed in range(6, 66):
    try:
        t.test_random_polarised_atmospheres_match_oracle(seed)
    except Exception as e:
        bad.append((seed, str(e).splitlines()[0][:200] if str(e) else type(e).__name__))
        print("FAIL seed", seed, bad[-1][1], flush=True)
print("polarised: failures", len(bad), "of 60")
bad2 = []
for seed in range(8, 48):
    try:
        t.test_random_spherical_atmospheres_match_oracle(seed)
    except Exception as e:
        bad2.append(seed); print("FAIL scalar seed", seed, str(e).splitlines()[0][:200] if str(e) else type(e).__name__, flush=True)
print("scalar: failures", len(bad2), "of 40")
