"""Small algorithm helpers with the reference's names (torchrl/algo/utils.py:5-32)."""
import torch


def huber(x, k=1.0):
    return torch.where(x.abs() < k, 0.5 * x.pow(2), k * (x.abs() - 0.5 * k))


def quantile_regression_loss(coefficient, source, target):
    diff = target.unsqueeze(-1) - source.unsqueeze(1)
    weight = (coefficient - (diff.detach() < 0).float()).abs()
    return (huber(diff) * weight).mean()


def soft_update_from_to(source, target, tau):
    with torch.no_grad():
        for tp, sp in zip(target.parameters(), source.parameters()):
            tp.data.mul_(1.0 - tau).add_(sp.data, alpha=tau)


def copy_model_params_from_to(source, target):
    with torch.no_grad():
        for tp, sp in zip(target.parameters(), source.parameters()):
            tp.data.copy_(sp.data)


def update_linear_schedule(optimizer, epoch, total_num_epochs, initial_lr):
    """lr = lr0 - lr0 * epoch / total (torchrl/algo/utils.py:28-32)."""
    lr = initial_lr - (initial_lr * (epoch / float(total_num_epochs)))
    for group in optimizer.param_groups:
        group['lr'] = lr
