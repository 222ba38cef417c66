import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from oracle import c_oracle
from tools import cpu_baselines as cb
text = open("pushworld_amd/data/puzzles/level4/Four Pistons.pwp").read()
pz = c_oracle.COraclePuzzle(text, order="cpp")
st = np.repeat(np.array([[x * 10000 + y for x, y in pz.initial_state]], np.int32), 1 << 20, axis=0)
out = c_oracle.expand4_batch(pz, st)
print("hw", cb.hardware_threads(), "cores", cb.physical_cores())
for th in (1, 8, 32, 64, 128, 256):
    cb.set_omp_threads(th)
    for pin in (False, True):
        ts = []
        for _ in range(3):
            if pin:
                with cb.pinned_threads():
                    t0 = time.perf_counter(); c_oracle.expand4_batch(pz, st, out); ts.append(time.perf_counter() - t0)
            else:
                t0 = time.perf_counter(); c_oracle.expand4_batch(pz, st, out); ts.append(time.perf_counter() - t0)
        print(th, "threads", "pinned" if pin else "free  ", ["%.1f ms" % (t * 1e3) for t in ts], "%.3e parents/s" % (len(st) / min(ts)), flush=True)
