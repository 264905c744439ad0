"""Hub model of the pre-training stages (reference model/pretrain/pr_hub_model.py:14-281): backbone + difference-map
decoder (masked modeling, `is_rec=True`) or backbone + MoCo-v3 heads + CLIP-token branch (contrastive stages).
Same constructor arguments, factories, forward signature, return tuples and state-dict keys as the reference."""
from functools import partial

import torch
import torch.nn as nn

from ... import ops
from ..backbone import convvit, swin, vit
from ..backbone.vit import init_linear_and_norm
from ..sub_module.mlp_head import _build_mlp_2d, run_mlp_2d
from . import pr_rec_decoder

_CON_PHASES = ("adj", "_adj", "con", "adj-n", "con-n", "rec+con")
_REC_PHASES = ("rec", "rec+con", "rec-n")


class PrHubModel(nn.Module):
    def __init__(self, args, patch_size=16, num_patches=196, embed_dim=1024, mlp_dim=4096,
                 proj_mlp_layers=3, pred_mlp_layers=2, norm_layer=nn.LayerNorm,
                 emb_frames_dim=512, queue_length=65536, T=0.07, rec_decoder_factory=None):
        super().__init__()
        self.args = args
        self.patch_size = patch_size
        self.T = T
        self.backbone_type = args.backbone_type
        self.mask_ratio = args.mask_ratio
        self.norm_pix_loss = args.norm_pix_loss
        common = dict(args=args, num_bins=args.num_bins, mask_ratio=args.mask_ratio, drop_rate=args.drop_rate,
                      attn_drop_rate=args.attn_drop_rate, drop_path_rate=args.drop_path_rate)
        if args.backbone_type == "vit":
            factory = {"small": "vit_small_patch16", "base": "vit_base_patch16", "tiny": "vit_tiny_patch16_64"}
            if args.model_size not in factory:
                raise ValueError(args.model_size)
            self.backbone = vit.__dict__[factory[args.model_size]](**common)
        elif args.backbone_type == "convvit":
            factory = {"small": "convvit_small_patch16", "base": "convvit_base_patch16"}
            if args.model_size not in factory:
                raise ValueError(args.model_size)
            self.backbone = convvit.__dict__[factory[args.model_size]](**common)
        elif args.backbone_type == "swin":
            factory = {"tiny": "swin_tiny_window7", "small": "swin_tiny_window7", "base": "swin_base_window7"}
            # ("small" keeps selecting Swin-T: the reference ignores model_size for swin, pr_hub_model.py:61-67)
            if args.model_size not in factory:
                raise ValueError(args.model_size)
            self.backbone = swin.__dict__[factory[args.model_size]](**common)
        else:
            raise ValueError(args.backbone_type)

        if args.pr_phase in _REC_PHASES:
            # The reference always builds the 384-wide "small" decoder here (pr_hub_model.py:77), which cannot take
            # a 768-wide backbone (SURVEY.md header). `rec_decoder_factory` lets the base / tiny factories pick the
            # decoder that fits; the default reproduces the reference.
            default = "pretrain_rec_decoder_swin_tiny_patch32" if args.backbone_type == "swin" else "pretrain_rec_decoder_small_patch16"
            name = rec_decoder_factory or default
            self.pretrain_rec_decoder = pr_rec_decoder.__dict__[name](frame_chans=args.frame_chans)

        if args.pr_phase in _CON_PHASES:
            C_out = embed_dim[-1]
            if args.use_queue:
                self.queue_length = queue_length
                q = torch.randn(C_out, num_patches, queue_length)
                self.register_buffer("queue", nn.functional.normalize(q, dim=0))
                self.register_buffer("queue_ptr", torch.zeros(1, dtype=torch.long))
            self.emb_h_proj = _build_mlp_2d(proj_mlp_layers, C_out, mlp_dim, C_out)
            self.emb_h_pred = _build_mlp_2d(pred_mlp_layers, C_out, mlp_dim, C_out)
            self.norm_clip_emb = norm_layer(emb_frames_dim)
            if args.backbone_type == "swin":          # 14x14 CLIP tokens -> the 7x7 grid of the last Swin stage
                self.clip_emb_proj = nn.Conv2d(emb_frames_dim, C_out, 2, stride=2)
            else:
                self.clip_emb_proj = nn.Linear(emb_frames_dim, C_out, bias=False)
        self.apply(init_linear_and_norm)

    # ------------------------------------------------------------------------------------------------ losses
    def reconstruct_loss(self, reconstruct_pred, sub_frame, mask):
        """Per-patch-normalised masked MSE against the patchified difference map (pr_hub_model.py:125-141)."""
        m = None if self.mask_ratio == 0 else mask.contiguous()
        return ops.RecLossFn.apply(reconstruct_pred, sub_frame, m, self.patch_size, self.norm_pix_loss)

    # ------------------------------------------------------------------------------------------------ queue under DDP
    def queue_policy(self):
        """How the MoCo queue is kept across data-parallel ranks (args.queue_policy):
          "all_gather" (default when args.distributed): every rank enqueues the keys of ALL ranks (contrastive-key all-gather
                        over xGMI), so the queues stay identical without any broadcast and hold world x B new keys per step;
          "rank0_broadcast": what the reference actually does -- DistributedDataParallel re-broadcasts module buffers from
                        rank 0 at every forward (main_pretrain.py:319, broadcast_buffers default), so every rank's queue,
                        pointer and BatchNorm statistics are overwritten by rank 0's before use and only rank 0's keys survive;
          "local": no communication (single process, or deliberately independent queues)."""
        pol = getattr(self.args, "queue_policy", None)
        if pol is None:
            pol = "all_gather" if (self.args.distributed and _dist_ready()) else "local"
        if pol not in ("all_gather", "rank0_broadcast", "local"):
            raise ValueError("queue_policy must be all_gather, rank0_broadcast or local")
        return pol if _dist_ready() else "local"

    _backward_cut = None

    def set_backward_cut(self, cut):
        """engine.BackwardCut (or None): the masked-modeling forward then hands the encoder output to the decoder through it, so
        that the decoder's backward and the encoder's backward can be run -- and captured -- as two calls. The data-parallel step
        executor uses this to start the all-reduce of the decoder's gradients under the encoder's backward."""
        self._backward_cut = cut

    _collective_hook = None

    def set_collective_hook(self, hook):
        """engine.ForwardCollectives (or None). While set, the forward hands its collectives to the hook instead of issuing them:
        the key all-gather of the in-batch InfoNCE splits the captured step ([forward to keys] -> all_gather_into_tensor into a
        static buffer -> [loss + backward]), the gathered enqueue of the MoCo queue and the reference-faithful buffer broadcast
        run outside the graphs -- so the data-parallel contrastive stage replays captured graphs like the other stages."""
        self._collective_hook = hook

    def forward_has_collective(self):
        """True when forward() talks to other ranks: the key all-gather of the queue / of the in-batch InfoNCE (pr_hub_model.py:
        248-259) or the reference-faithful buffer broadcast. The step executor keeps collectives outside captured HIP graphs:
        it captures such a forward through set_collective_hook (engine.ForwardCollectives)."""
        if self.args.pr_phase not in _CON_PHASES or not _dist_ready():
            return False
        pol = self.queue_policy()
        if pol == "rank0_broadcast":
            return True
        return pol == "all_gather" if self.args.use_queue else bool(self.args.distributed)

    @torch.no_grad()
    def _sync_buffers_from_rank0(self):
        """The reference-faithful mode's per-forward buffer broadcast (collective C2 in SURVEY.md 2.2)."""
        import torch.distributed as dist
        for b in self.buffers():
            dist.broadcast(b, src=0)

    @torch.no_grad()
    def _dequeue_and_enqueue(self, keys):
        """queue[:, :, ptr:ptr+B] = keys^T with all three dims reversed ((B,L,C) -> (C,L,B)), pointer advances
        modulo the queue length (pr_hub_model.py:112-122). The pointer stays in its device buffer: the kernel reads it
        and a one-thread kernel advances it, so the step has no host read-back (and can be captured in a HIP graph).
        Under the "all_gather" policy `keys` are first gathered from every rank (rank order), so B is world x local B."""
        if self.queue_policy() == "all_gather":
            import torch.distributed as dist
            if self.queue_length % (keys.shape[0] * dist.get_world_size()):
                raise AssertionError("queue_length must be a multiple of the (gathered) batch size")
            if self._collective_hook is not None:
                # captured data-parallel step: the all-gather and the enqueue leave the graph; the executor issues the gather right
                # after the forward graph (it crosses xGMI under the backward) and launches the enqueue behind it
                self._collective_hook.gather(keys, then=lambda gathered: ops.enqueue_keys_dev(self.queue, gathered, self.queue_ptr))
                return
            keys = concat_all_gather(keys)
        B = keys.shape[0]
        if self.queue_length % B:
            raise AssertionError("queue_length must be a multiple of the (gathered) batch size")
        ops.enqueue_keys_dev(self.queue, keys, self.queue_ptr)

    def contrastive_loss_queue(self, emb_h, clip_emb):
        loss, k = ops.info_nce_queue(emb_h, clip_emb, self.queue, self.T)
        self._dequeue_and_enqueue(k)
        return loss

    def contrastive_loss(self, emb_h, clip_emb):
        hook = self._collective_hook
        return ops.info_nce_inbatch(emb_h, clip_emb, self.T, distributed=bool(self.args.distributed),
                                    gather=None if hook is None else hook.gather)

    # ------------------------------------------------------------------------------------------------ forward
    def forward(self, events_voxel_grid, supp_data, is_rec=False, noise=None):
        swin_ = self.backbone_type == "swin"
        if is_rec and swin_:
            (emb_l1, emb_l2, emb_l3, emb_l4, emb_lh, coords_l1, coords_l2, coords_l3, coords_l4, mask, ids_restore,
             attn) = self.backbone(events_voxel_grid, mask=True, noise=noise)
            if self._backward_cut is not None:
                emb_lh = self._backward_cut(emb_lh)
            reconstruct_pred = self.pretrain_rec_decoder(emb_lh, ids_restore)
            reconstruct_loss = self.reconstruct_loss(reconstruct_pred, supp_data, mask)
            return (reconstruct_loss, emb_l1, emb_l2, emb_l3, emb_l4, emb_lh, coords_l1, coords_l2, coords_l3, coords_l4,
                    reconstruct_pred, mask, ids_restore, attn)
        if is_rec:
            emb_l1, emb_l2, emb_lh, mask, ids_restore = self.backbone(events_voxel_grid, mask=True, noise=noise)
            if self._backward_cut is not None:          # data-parallel step executor: decoder backward and encoder backward apart
                emb_lh = self._backward_cut(emb_lh)
            reconstruct_pred = self.pretrain_rec_decoder(emb_lh, ids_restore)
            reconstruct_loss = self.reconstruct_loss(reconstruct_pred, supp_data, mask)
            return reconstruct_loss, emb_l1, emb_l2, emb_lh, reconstruct_pred, mask, ids_restore

        if self.queue_policy() == "rank0_broadcast":
            if self._collective_hook is not None:
                self._collective_hook.pre_forward(self._sync_buffers_from_rank0)
            else:
                self._sync_buffers_from_rank0()
        if swin_:
            _, _, _, _, emb_h, attn = self.backbone(events_voxel_grid)
        else:
            _, _, emb_h, attn = self.backbone(events_voxel_grid)
        emb_h_org = emb_h.detach().clone()
        clip_emb = ops.LayerNormFn.apply(supp_data[:, 1:, :], None, None, self.norm_clip_emb.weight,
                                         self.norm_clip_emb.bias, self.norm_clip_emb.eps)
        clip_emb_org = clip_emb.detach().clone()
        if swin_:
            g = int(round(clip_emb.shape[1] ** 0.5))
            clip_emb_proj = ops.StridedConvTokensFn.apply(clip_emb, None, self.clip_emb_proj.weight, self.clip_emb_proj.bias,
                                                          2, g, g)
        else:
            clip_emb_proj = ops.LinearFn.apply(clip_emb, self.clip_emb_proj.weight, None)
        emb_h = run_mlp_2d(self.emb_h_proj, emb_h)
        emb_h_proj = run_mlp_2d(self.emb_h_pred, emb_h)
        if self.args.use_queue:
            contrastive_loss = self.contrastive_loss_queue(emb_h_proj, clip_emb_proj)
        else:
            contrastive_loss = self.contrastive_loss(emb_h_proj, clip_emb_proj)
        if self._backward_cut is not None:
            # data-parallel step executor: [forward] and [backward] as two captured graphs, so that the key all-gather issued
            # between them runs under the backward
            contrastive_loss = self._backward_cut(contrastive_loss)
        return contrastive_loss, emb_h_org, emb_h_proj, clip_emb_org, clip_emb_proj, attn


def _dist_ready():
    import torch.distributed as dist
    return dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1


@torch.no_grad()
def concat_all_gather(tensor):
    """all_gather along dim 0 without gradient (pr_hub_model.py:248-259) -- RCCL over xGMI under backend 'nccl'."""
    import torch.distributed as dist
    world = dist.get_world_size()
    out = torch.empty((world * tensor.shape[0],) + tuple(tensor.shape[1:]), dtype=tensor.dtype, device=tensor.device)
    dist.all_gather_into_tensor(out, tensor.contiguous())
    return out


def pretrain_hub_model_small_patch16(args, **kwargs):
    return PrHubModel(args=args, patch_size=16, num_patches=196, embed_dim=[128, 256, 384], mlp_dim=4096,
                      proj_mlp_layers=3, pred_mlp_layers=2, norm_layer=nn.LayerNorm, **kwargs)


def pretrain_hub_model_swin_tiny_patch16(args, **kwargs):
    """Swin-T hub (BASELINE.json config 5): 49 decoder cells of 32x32 pixels (pr_hub_model.py:269-274)."""
    return PrHubModel(args=args, patch_size=32, num_patches=49, embed_dim=[96, 192, 384, 768], mlp_dim=4096,
                      proj_mlp_layers=3, pred_mlp_layers=2, norm_layer=nn.LayerNorm, **kwargs)


def pretrain_hub_model_swin_base_patch16(args, **kwargs):
    """Swin-Base hub (BASELINE.json config 5 at its named size; needs args.model_size == "base"). No counterpart in the
    reference, which only builds Swin-T; same composition one size up."""
    kwargs.setdefault("rec_decoder_factory", "pretrain_rec_decoder_swin_base_patch32")
    return PrHubModel(args=args, patch_size=32, num_patches=49, embed_dim=[128, 256, 512, 1024], mlp_dim=4096,
                      proj_mlp_layers=3, pred_mlp_layers=2, norm_layer=nn.LayerNorm, **kwargs)


def pretrain_hub_model_base_patch16(args, **kwargs):
    """ViT-Base hub. The reference's factory of this name builds the 384-wide decoder and cannot run
    (SURVEY.md header); here the base decoder (pr_rec_decoder.py:89-95) is attached, which is the composition
    BASELINE.json config 2 names."""
    kwargs.setdefault("rec_decoder_factory", "pretrain_rec_decoder_base_patch16")
    return PrHubModel(args=args, patch_size=16, num_patches=196, embed_dim=[256, 384, 768], mlp_dim=4096,
                      proj_mlp_layers=3, pred_mlp_layers=2, norm_layer=nn.LayerNorm, **kwargs)


def pretrain_hub_model_tiny_patch16_64(args, **kwargs):
    """BASELINE.json config 1 plumbing model (64x64 voxels, 16 patches)."""
    kwargs.setdefault("rec_decoder_factory", "pretrain_rec_decoder_tiny_patch16_64")
    return PrHubModel(args=args, patch_size=16, num_patches=16, embed_dim=[192], mlp_dim=512,
                      proj_mlp_layers=3, pred_mlp_layers=2, norm_layer=nn.LayerNorm, **kwargs)
