"""Online meta-learning step of Meta-ViterbiNet: counterpart of Trainer.meta_train_loop (python_code/trainers/trainer.py:425-453)
with METAVNETTrainer.calc_loss (trainers/META_VNET/metavnet_trainer.py:41-50) and copy_model (utils/python_utils.py:17-27).
Plain PyTorch autograd (second-order through META_VNETDetector's functional forward); the Adam update goes through the
same optimizer state the one-launch online-training kernel uses (OnlineTrainer), like the single optimizer of the reference."""
import torch
from torch.nn import functional as F

from .trellis import calculate_states


def copy_model(source_model: torch.nn.Module, dest_model: torch.nn.Module):
    """Copy all parameters in place (python_utils.py:17-27)."""
    for s, d in zip(source_model.parameters(), dest_model.parameters()):
        d.data[:] = s.data[:]


def states_loss(soft_estimation: torch.Tensor, transmitted_words: torch.Tensor, memory_length: int) -> torch.Tensor:
    """CrossEntropy(mean) between the logits of every symbol and its trellis state (metavnet_trainer.py:41-50)."""
    gt_states = calculate_states(memory_length, transmitted_words)
    return F.cross_entropy(soft_estimation.reshape(-1, soft_estimation.shape[-1]), gt_states)


def meta_train_loop(detector, meta_detector, online_trainer, received_words: torch.Tensor, transmitted_words: torch.Tensor,
                    support_idx: torch.Tensor, query_idx: torch.Tensor, meta_lr: float, MAML: bool = True) -> torch.Tensor:
    """One MAML step (trainer.py:425-453): inner SGD step on the support words, query loss through the updated weights,
    meta-gradient w.r.t. the original weights (second order when MAML), one Adam step on the detector."""
    support_tx, support_rx = transmitted_words[support_idx], received_words[support_idx]
    query_tx, query_rx = transmitted_words[query_idx], received_words[query_idx]
    params = list(detector.parameters())
    soft_supp = meta_detector(support_rx, "train", params)
    loss_supp = states_loss(soft_supp, support_tx, online_trainer.memory_length)
    local_grad = torch.autograd.grad(loss_supp, params, create_graph=MAML)
    updated = [p - meta_lr * g for g, p in zip(local_grad, params)]
    soft_query = meta_detector(query_rx, "train", updated)
    loss_query = states_loss(soft_query, query_tx, online_trainer.memory_length)
    meta_grad = torch.autograd.grad(loss_query, params, create_graph=False)
    online_trainer.adam_step(meta_grad)
    return loss_query.detach()
