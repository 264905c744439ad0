"""Classification fine-tuning hub (reference model/finetune_cls/ft_cls_hub_model.py:6-152): dense backbone branch ->
mean over the tokens -> Linear head. Same constructor, factories, forward return tuples and state-dict keys
(`backbone.*`, `classify_head.{weight,bias}`), so a pre-trained `backbone.*` checkpoint loads with strict=False as in
main_finetune_cls.py. Backbones: vit / convvit / swin (the ECDP / MEM / ECDDP baselines are out of scope)."""
import torch.nn as nn

from ... import ops
from ..backbone import convvit, swin, vit
from ..backbone.vit import init_linear_and_norm


class FtClsHubModel(nn.Module):
    def __init__(self, args, embed_dim=1024):
        super().__init__()
        self.backbone_type = args.backbone_type
        common = dict(args=args, num_bins=args.num_bins, drop_rate=args.drop_rate, attn_drop_rate=args.attn_drop_rate,
                      drop_path_rate=args.drop_path_rate)
        if args.backbone_type == "vit":
            factory = {"small": "vit_small_patch16", "base": "vit_base_patch16"}
            if args.model_size not in factory:
                raise ValueError(args.model_size)
            self.backbone = vit.__dict__[factory[args.model_size]](**common)
        elif args.backbone_type == "convvit":
            factory = {"small": "convvit_small_patch16", "base": "convvit_base_patch16"}
            if args.model_size not in factory:
                raise ValueError(args.model_size)
            self.backbone = convvit.__dict__[factory[args.model_size]](**common)
        elif args.backbone_type == "swin":
            self.backbone = swin.__dict__["swin_tiny_window7"](**common)
        elif args.backbone_type in ("vit_ecdp", "convvit_ecdp", "vit_mem", "swin_ecddp"):
            raise NotImplementedError("%s is one of the reference's comparison baselines (out of scope)" % args.backbone_type)
        else:
            raise ValueError(args.backbone_type)
        self.classify_head = nn.Linear(embed_dim[-1], args.num_classes)
        self.apply(init_linear_and_norm)

    def forward(self, x):
        if self.backbone_type == "swin":
            emb_l1, emb_l2, emb_l3, emb_l4, emb_h, attn = self.backbone(x)
        else:
            emb_l1, emb_l2, emb_h, attn = self.backbone(x)
        emb_h_pool = ops.TokenMeanFn.apply(emb_h)                      # global pool without cls token
        pred = ops.LinearFn.apply(emb_h_pool, self.classify_head.weight, self.classify_head.bias)
        if self.backbone_type == "swin":
            return emb_l1, emb_l2, emb_l3, emb_l4, emb_h, pred, attn
        return emb_l1, emb_l2, emb_h, pred, attn


def finetune_cls_hub_model_small_patch16(args):
    return FtClsHubModel(args=args, embed_dim=[128, 256, 384])


def finetune_cls_hub_model_swin_tiny_window7(args):
    return FtClsHubModel(args=args, embed_dim=[96, 192, 384, 768])


def finetune_cls_hub_model_base_patch16(args):
    return FtClsHubModel(args=args, embed_dim=[256, 384, 768])
