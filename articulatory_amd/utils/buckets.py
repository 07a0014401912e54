"""Bucketed, overlapped gradient all-reduce for data-parallel training (one process per GPU, RCCL over xGMI under torch.distributed's
"nccl" backend).  The reference meant to wrap both networks in DistributedDataParallel (articulatory/bin/train.py:1790-1801), which
all-reduces buckets of gradients while backward still runs; libhificar's backward passes are single native calls, so they report the
moment a bucket's gradients are complete through a host callback (include/hificar.h: hificar_bucket_fn) and this module starts that
bucket's collective right there: ``reduce(bucket)`` = ``all_reduce(async_op=True)`` on the bucket's slices of the flat raw-gradient
buffer — the communication stream waits for the work enqueued so far on the CURRENT stream only, and runs beside the rest of the backward
pass; ``finish()`` waits for every collective (making the current stream wait, not the host) and averages like DistributedDataParallel.

Pure torch.distributed logic (no native code): tests/test_distributed_gloo.py runs it at world size 2 on CPU tensors over gloo and checks
it against ONE all-reduce of the whole buffer."""
import torch.distributed as dist


def bucket_ranges(raw_buckets, numels, n_buckets):
    """raw_buckets[i] / numels[i]: bucket and element count of raw parameter i; the raw-gradient buffer holds the parameters back to back,
    every slot rounded up to 4 floats.  -> ([bucket] list of (offset, length) contiguous float ranges — adjacent slots of one bucket
    merged —, total floats)."""
    ranges = [[] for _ in range(n_buckets)]
    off = 0
    for b, n in zip(raw_buckets, numels):
        n4 = (int(n) + 3) & ~3
        r = ranges[b]
        if r and r[-1][0] + r[-1][1] == off:
            r[-1] = (r[-1][0], r[-1][1] + n4)
        else:
            r.append((off, n4))
        off += n4
    return ranges, off


class BucketReducer:
    """All-reduce a flat gradient buffer bucket by bucket, as the buckets complete."""

    def __init__(self, raw, ranges, group=None, average=True):
        self.raw, self.ranges, self.group, self.average = raw, ranges, group, average
        self.pending, self.done = [], set()

    def reduce(self, bucket):
        """Start the collective(s) of one bucket; call on the stream (torch.cuda.stream context) that produced its gradients."""
        if bucket in self.done:
            raise RuntimeError(f"bucket {bucket} reduced twice")
        self.done.add(bucket)
        for off, n in self.ranges[bucket]:
            self.pending.append(dist.all_reduce(self.raw[off:off + n], group=self.group, async_op=True))

    def finish(self):
        """Every bucket must have been started; waits for all of them and averages.  Returns the buffer."""
        missing = [b for b in range(len(self.ranges)) if b not in self.done and self.ranges[b]]
        if missing:
            raise RuntimeError(f"buckets {missing} were never reduced")
        for w in self.pending:
            w.wait()
        self.pending = []
        if self.average:
            self.raw.div_(dist.get_world_size(self.group))
        return self.raw


class BucketHook:
    """Host side of ``hificar_bucket_fn`` (include/hificar.h) for one backward pass: libhificar calls it — on the host, from inside the native
    backward — right after bucket ``b``'s last gradient kernel has been enqueued on ``bstream``.  The hook runs the bucket's weight-norm chain rule
    behind those kernels (``chain_rule(bucket, bstream)``) and starts the bucket's all-reduce on that stream (``stream_ctx(bstream)`` makes it the
    current one: the collective waits for what is enqueued on it so far and runs beside the rest of the backward pass).  An exception must not
    cross the C frames: it is kept and re-raised by ``finish()``, which otherwise waits for every collective and averages.

    Every rank's native backward reports its buckets in the same order (same code, same shapes), which is what the collectives need; the order is
    NOT ascending and the callbacks of the discriminators come from eight different streams."""

    def __init__(self, reducer, chain_rule, stream_ctx=None):
        self.reducer, self.chain_rule, self.stream_ctx, self.errors = reducer, chain_rule, stream_ctx, []

    def __call__(self, bucket, bstream, _user=None):
        try:
            if self.stream_ctx is None:
                self.chain_rule(bucket, bstream)
                self.reducer.reduce(bucket)
            else:
                with self.stream_ctx(bstream):
                    self.chain_rule(bucket, bstream)
                    self.reducer.reduce(bucket)
        except BaseException as e:  # noqa: BLE001  (re-raised on the Python side of the call)
            self.errors.append(e)

    def finish(self):
        if self.errors:
            for w in self.reducer.pending:  # collectives already started must still complete on every rank
                w.wait()
            raise self.errors[0]
        return self.reducer.finish()
