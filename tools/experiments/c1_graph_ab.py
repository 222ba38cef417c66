"""C1 (gym step with the default float32 ppc-20 observation): eager launches against one graph replay per step
(PUSHWORLD_AMD_STEP_GRAPHS), same process order alternated; best of several segments."""
import os, subprocess, sys, tempfile, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

def run():
    from pushworld_amd import benchmark_data as bd
    from pushworld_amd.gym_env import PushWorldEnv
    member, text = next(iter(bd.level0_texts(("base",), "train", 1).items()))
    acts = np.random.default_rng(0).integers(0, 4, 10000)
    with tempfile.TemporaryDirectory() as tmp:
        path = os.path.join(tmp, os.path.basename(member))
        open(path, "w").write(text)
        env = PushWorldEnv(path, max_steps=100)
        env.reset(seed=0)
        for a in acts[:200]:
            _, _, te, tr, _ = env.step(int(a))
            if te or tr:
                env.reset()
        rates = []
        for seg in range(5):
            t0 = time.perf_counter()
            for a in acts[seg * 1500:(seg + 1) * 1500]:
                _, _, te, tr, _ = env.step(int(a))
                if te or tr:
                    env.reset()
            rates.append(1500 / (time.perf_counter() - t0))
        print(os.environ.get("PUSHWORLD_AMD_STEP_GRAPHS", "1"), "graphs" if env._graphs else "eager", [round(r) for r in rates], flush=True)

if __name__ == "__main__":
    if len(sys.argv) > 1:
        run()
    else:
        for v in ("0", "1", "0", "1"):
            subprocess.run([sys.executable, __file__, "x"], env=dict(os.environ, PUSHWORLD_AMD_STEP_GRAPHS=v))
