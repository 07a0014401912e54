"""Per-rank host resources for one-process-per-GPU runs (``torchrun --nproc-per-node N``).

The training iteration enqueues ~1000 launches from Python per step and sits close to host-bound on one GPU; N ranks that
all spread their OpenMP / intra-op pools over every core of the host, or migrate between sockets, would make the host the bottleneck of an
8-GPU run.  ``pin_rank`` gives rank r of W local ranks the r-th contiguous W-th of the cores this process may run on (``sched_setaffinity``) and
sizes torch's intra-op pool to it.  Call it BEFORE ``init_process_group`` and before any parallel torch work (bench.py and bin/train.py do): the
mask is applied to every thread the process has at that moment, later threads inherit it.  The reference's recipes export OMP_NUM_THREADS=1 instead (egs/ema/voc1/path.sh:13).  HIFICAR_NO_AFFINITY=1
turns it off.
"""

import os


def pin_rank(local_rank, local_world, max_threads=16):
    """-> a short description of what was done (for logs / the bench line), or None when nothing was changed."""
    if local_world <= 1 or os.environ.get("HIFICAR_NO_AFFINITY") == "1" or not hasattr(os, "sched_getaffinity"):
        return None
    cores = sorted(os.sched_getaffinity(0))
    per = len(cores) // local_world
    if per < 1:
        return None
    mine = cores[local_rank * per:(local_rank + 1) * per]
    os.sched_setaffinity(0, mine)
    # sched_setaffinity(0) moves the CALLING thread only; threads that already exist (an OpenMP pool, a pin-memory or watchdog thread of a
    # library imported earlier) keep the full mask unless they are moved too.  Threads created from here on inherit the slice.
    try:
        for tid in os.listdir("/proc/self/task"):
            try:
                os.sched_setaffinity(int(tid), mine)
            except OSError:  # (a thread that exited meanwhile)
                pass
    except OSError:  # pragma: no cover
        pass
    n = max(1, min(per, max_threads))
    os.environ["OMP_NUM_THREADS"] = str(n)  # (for libraries initialised after this point)
    try:
        import torch

        torch.set_num_threads(n)
    except Exception:  # pragma: no cover
        pass
    return f"cores {mine[0]}-{mine[-1]} ({per} of {len(cores)}), {n} intra-op threads"
