import os, sys, time, torch
sys.path.insert(0, "oracle")
import mtn_oracle as O
print("cpu_count", os.cpu_count(), "affinity", len(os.sched_getaffinity(0)))
sd = O.synthetic_state_dict(21, 19, seed=0)
batch = O.synthetic_batch(16, 192, seed=1234, gaussian_targets=False)
for th in (8, 16, 32, 64):
    torch.set_num_threads(th)
    st = {}
    t0 = time.perf_counter(); O.train_step(sd, batch, st, 1e-3); t1 = time.perf_counter()
    O.train_step(sd, batch, st, 1e-3); t2 = time.perf_counter()
    print(th, "threads: first %.2fs second %.2fs -> %.1f img/s" % (t1 - t0, t2 - t1, 16 / (t2 - t1)), flush=True)
