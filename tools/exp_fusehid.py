"""Experiment: cfg2 fp32 step with hidden produced inside the forward kernel (default) vs by the
separate k_make_hidden pass (rnnt_engine_set_flags(64))."""
import sys, time
sys.path.insert(0, ".")
import torch
from rnnt_amd import engine

if __name__ == "__main__":
    import bench
    for flags in (0, 64, 0, 64):
        engine.lib().rnnt_engine_set_flags(flags)
        sys.argv = ["bench.py", "--steps", "5", "--warmup", "2", "--no-cpu-baseline"]
        t0 = time.time()
        bench.main()
        print("flags", flags, "wall", round(time.time() - t0, 1), flush=True)
    engine.lib().rnnt_engine_set_flags(0)
