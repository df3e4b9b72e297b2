"""Online meta-learning step of Meta-ViterbiNet: counterpart of Trainer.meta_train_loop (python_code/trainers/trainer.py:425-453)
with METAVNETTrainer.calc_loss (trainers/META_VNET/metavnet_trainer.py:41-50) and copy_model (utils/python_utils.py:17-27).
Plain PyTorch autograd (second-order through META_VNETDetector's functional forward); the Adam update goes through the
same optimizer state the one-launch online-training kernel uses (OnlineTrainer), like the single optimizer of the reference."""
import torch
from torch.nn import functional as F

from .trellis import calculate_states


def copy_model(source_model: torch.nn.Module, dest_model: torch.nn.Module):
    """Copy all parameters in place (python_utils.py:17-27); one multi-tensor copy instead of a launch per parameter."""
    # (a detector keeps the list of its Parameter objects, detectors.VNETDetector._params: walking the module tree twice per call
    # costs more than the copy)
    sp = source_model._params() if hasattr(source_model, "_params") else source_model.parameters()
    dp = dest_model._params() if hasattr(dest_model, "_params") else dest_model.parameters()
    src, dst = [p.data for p in sp], [p.data for p in dp]
    if src and all(a.shape == b.shape for a, b in zip(src, dst)) and len(src) == len(dst):
        torch._foreach_copy_(dst, src)
    else:  # (shapes that differ: let the element-wise assignment raise what the reference's would)
        for a, b in zip(src, dst):
            b[:] = a[:]


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
    online_trainer.optimizer_step(meta_grad)  # the trainer's single optimizer (trainer.py:452)
    return loss_query.detach()


class GraphedMetaStep:
    """meta_train_loop as ONE hipGraph replay (torch.cuda.CUDAGraph): the ~200 small ATen kernels of a second-order MAML
    step (forward, double backward, Adam) are captured once for a given (support words, word length) shape and replayed
    per step with the (support, query) words copied into static buffers.  Same kernels in the same order as the eager
    function, so the arithmetic is the same up to the form of the Adam update (OnlineTrainer.adam_step_device)."""

    def __init__(self, detector, meta_detector, online_trainer, n_support: int, T: int, meta_lr: float, MAML: bool = True):
        self.detector, self.meta_detector, self.tr = detector, meta_detector, online_trainer
        self.meta_lr, self.MAML = meta_lr, MAML
        params = list(detector.parameters())
        dev = params[0].device
        if dev.type != "cuda":
            raise ValueError("GraphedMetaStep needs the detector on the GPU")
        self.s_rx, self.s_tx = torch.zeros(n_support, T, device=dev), torch.zeros(n_support, T, device=dev)
        self.q_rx, self.q_tx = torch.zeros(1, T, device=dev), torch.zeros(1, T, device=dev)
        self.step_t = torch.zeros((), dtype=torch.float64, device=dev)
        self.loss = torch.zeros((), device=dev)
        # the warm-up passes torch asks for before a capture execute for real: put the state back afterwards
        snap = [p.detach().clone() for p in params] + [online_trainer.exp_avg.clone(), online_trainer.exp_avg_sq.clone()]
        side = torch.cuda.Stream(dev)
        side.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(side):
            for _ in range(2):
                self._body()
        torch.cuda.current_stream(dev).wait_stream(side)
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph):
            self._body()
        with torch.no_grad():
            for p, a in zip(params, snap):
                p.copy_(a)
            online_trainer.exp_avg.copy_(snap[-2])
            online_trainer.exp_avg_sq.copy_(snap[-1])

    def _body(self):
        with torch.enable_grad():
            params = list(self.detector.parameters())
            soft_supp = self.meta_detector(self.s_rx, "train", params)
            loss_supp = states_loss(soft_supp, self.s_tx, self.tr.memory_length)
            local_grad = torch.autograd.grad(loss_supp, params, create_graph=self.MAML)
            updated = [p - self.meta_lr * g for g, p in zip(local_grad, params)]
            soft_query = self.meta_detector(self.q_rx, "train", updated)
            loss_query = states_loss(soft_query, self.q_tx, self.tr.memory_length)
            meta_grad = torch.autograd.grad(loss_query, params, create_graph=False)
        self.tr.adam_step_device(meta_grad, self.step_t)
        self.loss.copy_(loss_query.detach())

    def __call__(self, received_words: torch.Tensor, transmitted_words: torch.Tensor, support_idx: torch.Tensor,
                 query_idx: torch.Tensor) -> torch.Tensor:
        """One MAML step on words picked from the buffers (indices may be negative, like the reference's)."""
        self.s_rx.copy_(received_words[support_idx])
        self.s_tx.copy_(transmitted_words[support_idx])
        self.q_rx.copy_(received_words[query_idx])
        self.q_tx.copy_(transmitted_words[query_idx])
        self.step_t.fill_(float(self.tr.step))
        self.graph.replay()
        self.tr.step += 1
        return self.loss
