"""Per-iteration learning-rate schedule: linear warm-up then half-cosine (reference utils/lr_sched.py:3-16)."""
import math


def adjust_learning_rate(optimizer, epoch, args):
    """`epoch` is fractional (iteration / iterations_per_epoch + epoch). Writes lr (times the group's optional
    `lr_scale`) into every param group and returns the unscaled lr."""
    if epoch < args.warmup_epochs:
        lr = args.lr * epoch / args.warmup_epochs
    else:
        progress = (epoch - args.warmup_epochs) / (args.epochs - args.warmup_epochs)
        lr = args.min_lr + (args.lr - args.min_lr) * 0.5 * (1.0 + math.cos(math.pi * progress))
    for group in optimizer.param_groups:
        group["lr"] = lr * group.get("lr_scale", 1.0)
    return lr
