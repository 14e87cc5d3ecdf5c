"""bench.py's workloads executed top to bottom on a machine WITHOUT a GPU (tools/dry_env.py's stubs: no-op launches,
dummy stream / graph / event objects, oracle-made scenes).  Under `python -m torch.distributed.run` the ranks and collectives
are real (gloo).  The numbers mean nothing; the point is that every line of the glue code runs -- timing loops, HIP-event
bookkeeping, roofline blocks, profile evidence, the parity block against the oracle, capture / replay bookkeeping, rank
agreement, JSON assembly -- before the driver runs it for real.  Prints the JSON line bench.py would print.

    python tools/bench_dry_run.py --points 2000 --steps 3 --warmup 1 --no-train
    python tools/bench_dry_run.py --workload train --model configPCF_2cm_PTF2 --points 1200 --scenes 2 --steps 2 --warmup 1"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, 'ml-pointconvformer_amd'), os.path.join(ROOT, 'tools')):
    if p not in sys.path:
        sys.path.insert(0, p)

import torch  # noqa: E402

import dry_env  # noqa: E402


def main():
    dry_env.install()
    import pcf_dist
    cpu = torch.device('cpu')
    if 'WORLD_SIZE' in os.environ:          # under torch.distributed.run: real ranks and real collectives, over gloo
        def setup(backend=None):
            rank, world, local_rank = pcf_dist.env_rank()
            if not torch.distributed.is_initialized():
                torch.distributed.init_process_group('gloo', rank=rank, world_size=world)
            return rank, world, local_rank, cpu
        pcf_dist.setup = setup
    else:
        pcf_dist.setup = lambda backend=None: (0, 1, 0, cpu)
        pcf_dist.fence = lambda dev=None: None
    import bench
    bench.main()


if __name__ == '__main__':
    main()
