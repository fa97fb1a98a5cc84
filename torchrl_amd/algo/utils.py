"""Host-side helpers under the reference's names (torchrl/algo/utils.py:5-32).  On the hot path their work is done by
kernels (trl_quantile_huber_f32, trl_polyak_f32, the flat-buffer copy of the target policy, the device learning
rates); these module-level versions serve the reference's call sites and the tests."""
import torch


def huber(x, k=1.0):
    """0.5 x^2 inside [-k, k], k (|x| - k / 2) outside."""
    magnitude = x.abs()
    return torch.where(magnitude < k, 0.5 * x.pow(2), k * (magnitude - 0.5 * k))


def quantile_regression_loss(coefficient, source, target):
    """mean over (batch, target quantile i, source quantile j) of huber(T_i - theta_j) * |tau_j - 1[T_i < theta_j]|."""
    pairwise = target.unsqueeze(-1) - source.unsqueeze(1)
    below = (pairwise.detach() < 0).float()
    return ((coefficient - below).abs() * huber(pairwise)).mean()


def _parameter_pairs(source, target):
    return zip(source.parameters(), target.parameters())


@torch.no_grad()
def soft_update_from_to(source, target, tau):
    """target <- (1 - tau) target + tau source, parameter by parameter."""
    for src, dst in _parameter_pairs(source, target):
        dst.data.mul_(1.0 - tau).add_(src.data, alpha=tau)


@torch.no_grad()
def copy_model_params_from_to(source, target):
    for src, dst in _parameter_pairs(source, target):
        dst.data.copy_(src.data)


def update_linear_schedule(optimizer, epoch, total_num_epochs, initial_lr):
    """Every param group gets lr0 - lr0 * epoch / total (utils.py:28-32; that exact expression, not lr0 * (1 - e/t))."""
    fraction = epoch / float(total_num_epochs)
    value = initial_lr - (initial_lr * fraction)
    for group in optimizer.param_groups:
        group['lr'] = value
