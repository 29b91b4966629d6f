"""Worker of tests/test_bench_launcher.py::test_ranks_stop_the_timed_region_together: bench.time_blocks over gloo with ranks whose
frames take different times -- every rank must run the same count of blocks (a rank that stopped on its own clock would leave the
others waiting in the next block's barrier)."""
import os
import sys
import time

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench    # noqa: E402


class Loop:
    def __init__(self, seconds):
        self.seconds = seconds

    def frame(self, i):
        time.sleep(self.seconds)


def main():
    out_path, collective = sys.argv[1], sys.argv[2] == "1"
    dist.init_process_group("gloo", timeout=__import__("datetime").timedelta(seconds=120))
    rank = dist.get_rank()

    def slowest(dt):
        t = torch.tensor([dt], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())
    # rank 0's frames last 4 ms, rank 1's 1 ms outside the barriers: rank 1 alone would want more blocks than rank 0 ... except that the
    # barriers make every block last as long as the slowest rank's; what differs is each rank's own clock around them, so the stop
    # threshold sits right at a block boundary to make the clocks disagree
    loop = Loop(0.004 if rank == 0 else 0.001)
    per_block = 2 * 0.004
    times, f = bench.time_blocks(loop, dist.barrier, 0, 2, 5 * per_block, max_blocks=50, slowest=slowest if collective else None)
    counts = [None, None]
    dist.all_gather_object(counts, len(times))
    if rank == 0:
        with open(out_path, "w") as fh:
            fh.write(" ".join(str(c) for c in counts))
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
