"""Parameter groups for AdamW (reference utils/lr_decay.py:16-106): weight decay only on parameters with more than
one dimension; optional layer-wise lr decay / "layer grafted" scales keyed by the block index in the name."""
import json


def _layer_id(name, num_layers, backbone_type, layer_grafted):
    emb = name.startswith("backbone.pos_embed") or name.startswith("backbone.patch_embed")
    if layer_grafted:
        if emb or name.startswith("backbone.conv_block1") or name.startswith("backbone.conv_block2"):
            return 0
        if name.startswith("backbone.vit_block"):
            return min(int(name.split(".")[2]) // 4, 2)
        return 2
    if emb:
        return 0
    if name.startswith("backbone.vit_block"):
        idx = int(name.split(".")[2])
        if backbone_type in ("vit", "vit_mem", "vit_ecdp"):
            return idx + 1
        if backbone_type == "convvit":
            return idx + 3
        return None
    if name.startswith("backbone.conv_block1"):
        return 1
    if name.startswith("backbone.conv_block2"):
        return 2
    return num_layers


def param_groups_lrd(args, model, weight_decay=0.05, no_weight_decay_list=(), layer_decay=0.75, layer_grafted=False):
    bt = args.backbone_type
    if bt in ("vit", "vit_mem", "vit_ecdp"):
        num_layers = len(model.backbone.vit_block)
    elif bt == "convvit":
        num_layers = len(model.backbone.vit_block) + 2
    elif bt == "swin":
        num_layers = len(model.backbone.swin_block)
    else:
        num_layers = 0
    scales = [0.01, 0.1, 1] if layer_grafted else [layer_decay ** (num_layers - i) for i in range(num_layers + 1)]
    groups, names = {}, {}
    for n, p in model.named_parameters():
        if not p.requires_grad:
            continue
        no_decay = p.ndim == 1 or n in no_weight_decay_list
        lid = _layer_id(n, num_layers, bt, layer_grafted)
        key = "layer_%d_%s" % (lid, "no_decay" if no_decay else "decay")
        if key not in groups:
            groups[key] = {"lr_scale": scales[lid], "weight_decay": 0.0 if no_decay else weight_decay, "params": []}
            names[key] = {"lr_scale": scales[lid], "weight_decay": groups[key]["weight_decay"], "params": []}
        groups[key]["params"].append(p)
        names[key]["params"].append(n)
    if layer_decay != 1 or layer_grafted:
        print("parameter groups: \n%s" % json.dumps(names, indent=2))
    return list(groups.values())
