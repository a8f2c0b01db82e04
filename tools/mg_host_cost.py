"""What one GPU can measure of the multi-GPU step: the driver at world size 1 with its collectives on
(RCCL all-reduce of the two scalars, in-place all-gather of the source vector), against the plain session.
Prints per iteration: host time to ENQUEUE the step (the Python loop of MGPagerank.iterate: no sync inside),
wall time of the step, and the same for the session alone.  usage: mg_host_cost.py [scale] [iterations]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.distributed as dist
from cugraph_amd import mg, pylibcugraph as plc

scale = int(sys.argv[1]) if len(sys.argv) > 1 else 26
n = int(sys.argv[2]) if len(sys.argv) > 2 else 50
os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29541")
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
h = plc.ResourceHandle(rank=0, comm_size=1)
drv = mg.MGPagerank.from_rmat(h, plc.GraphProperties(is_multigraph=True), scale, 16 << scale, seed=42)
for always in (False, True):
    drv.always_exchange = always
    sess = drv.session(alpha=0.85, epsilon=0.0)
    drv.iterate(sess, 5)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    drv.iterate(sess, n)
    t_host = time.perf_counter() - t0
    torch.cuda.synchronize()
    t_wall = time.perf_counter() - t0
    print(f"scale {scale} collectives {'on ' if always else 'off'}: host enqueue {t_host / n * 1e6:7.1f} us/iteration, "
          f"wall {t_wall / n * 1e3:.3f} ms/iteration", flush=True)
    del sess
dist.destroy_process_group()
