"""Data-parallel gradient reduction for one process per GPU: RCCL (torch.distributed backend "nccl" on ROCm) over
xGMI. Replaces the reference's DistributedDataParallel wrap (main_pretrain.py:319; collective C1 in SURVEY.md 2.2).

Gradients are packed into a few large flat f32 buckets in reverse parameter order (the order backward produces
them); each bucket's SUM all-reduce is issued asynchronously from a post-accumulate-grad hook as soon as its last
gradient has landed, so the collective overlaps the rest of backward. Buckets are large (default 64 MB): xGMI is
point-to-point and a ring step is per-link bound, so few big messages beat many 25 MB ones. The 1/world mean is
folded into FusedAdamW's grad_scale (no extra pass). Buffers (MoCo queue, BN statistics) are NOT broadcast each
forward -- the reference's accidental C2 broadcast is deliberately not reproduced.
"""
import torch
import torch.distributed as dist


class BucketedGradReducer:
    def __init__(self, params, bucket_mb=64.0, process_group=None):
        self.group = process_group
        self.params = [p for p in params if p.requires_grad]
        cap = int(bucket_mb * 1024 * 1024 / 4)
        self.buckets = []          # dict(buf, params=[(p, off, n)], pending, work)
        cur, cur_n = [], 0
        for p in reversed(self.params):
            n = p.numel()
            if cur and cur_n + n > cap:
                self._close(cur, cur_n)
                cur, cur_n = [], 0
            cur.append((p, cur_n, n))
            cur_n += (n + 63) // 64 * 64          # keep every slice 256-byte aligned
        if cur:
            self._close(cur, cur_n)
        self._where = {}
        for bi, b in enumerate(self.buckets):
            for (p, off, n) in b["params"]:
                self._where[p] = (bi, off, n)
                p.register_post_accumulate_grad_hook(self._hook)
        self._reset()

    def _close(self, items, total):
        dev = items[0][0].device
        self.buckets.append(dict(buf=torch.zeros(total, dtype=torch.float32, device=dev), params=list(items), work=None))

    def _reset(self):
        for b in self.buckets:
            b["pending"] = len(b["params"])
            b["work"] = None

    def _hook(self, p):
        bi, off, n = self._where[p]
        b = self.buckets[bi]
        view = b["buf"][off:off + n].view_as(p)
        view.copy_(p.grad)                 # D2D copy into the bucket; the optimizer then reads the reduced view
        p.grad = view
        b["pending"] -= 1
        if b["pending"] == 0:
            b["work"] = dist.all_reduce(b["buf"], op=dist.ReduceOp.SUM, group=self.group, async_op=True)

    def finish(self):
        """Call after backward: flush buckets whose parameters got no gradient this step, wait for all collectives
        (on the current stream, no host sync), re-arm."""
        for b in self.buckets:
            if b["work"] is None and b["pending"] < len(b["params"]):
                b["work"] = dist.all_reduce(b["buf"], op=dist.ReduceOp.SUM, group=self.group, async_op=True)
        for b in self.buckets:
            if b["work"] is not None:
                b["work"].wait()
        self._reset()
