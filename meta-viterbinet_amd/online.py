"""Online (self-supervised) training of the ViterbiNet MLP in ONE kernel launch per word: counterpart of
VNETTrainer.online_training (python_code/trainers/VNET/vnet_trainer.py:49-60) / run_train_loop (trainers/trainer.py:492-505)
with CrossEntropyLoss + Adam, the step that dominates the by-word evaluation with self_supervised=True."""
import torch

from . import _lib
from .trellis import calculate_states


class OnlineTrainer:
    """Holds the Adam state (exp_avg, exp_avg_sq, step) for a VNETDetector's six parameters, like the optimizer
    deep_learning_setup() creates (trainer.py:163-173), and runs `iterations` CE+Adam steps on one word on the GPU."""

    def __init__(self, detector, memory_length: int, lr: float = 0.001, betas=(0.9, 0.999), eps: float = 1e-8,
                 train_minibatch_size: int = 32, use_kernel: bool = True, optimizer_type: str = "Adam"):
        if optimizer_type not in ("Adam", "RMSprop", "SGD"):  # deep_learning_setup (trainer.py:163-175)
            raise NotImplementedError("No such optimizer implemented!!!")
        self.detector = detector
        self.optimizer_type = optimizer_type
        # False: online_training on stock PyTorch autograd (the cross-check of the kernel).  The one-launch online-training kernels
        # implement all three optimizers of deep_learning_setup (Adam, the reference's default, config.yaml:35; RMSprop and SGD with
        # torch's defaults); the meta-learning kernel implements Adam (harness.eval_by_word takes autograd for the others).
        self.use_kernel = use_kernel
        self.memory_length = memory_length
        self.lr, self.betas, self.eps = lr, betas, eps
        self.train_minibatch_size = train_minibatch_size
        self.params = list(detector.net.parameters())
        dev = self.params[0].device
        n = sum(p.numel() for p in self.params)
        self.exp_avg = torch.zeros(n, dtype=torch.float32, device=dev)
        self.exp_avg_sq = torch.zeros(n, dtype=torch.float32, device=dev)
        self.step = 0
        # set to 1 by a training launch whose device-wide barrier gave up (include/mvn.h); read by check_status().  It is the
        # second of two words: the first is free for the caller's per-block error count (harness.eval_by_word), so that ONE
        # device-to-host copy per block brings both (sync_words)
        self.sync_words = torch.zeros(2, dtype=torch.int32, device=dev) if dev.type == "cuda" else None
        self.status = self.sync_words[1:2] if dev.type == "cuda" else None
        self._unchecked = False

    def kernel_optimizer_args(self):
        """(beta1, beta2, eps) as the training kernels take them: Adam's own; beta1 = MVN_BETA1_RMSPROP (-1) with alpha = 0.99 and
        eps = 1e-8 (torch.optim.RMSprop's defaults, as deep_learning_setup builds it); beta1 = MVN_BETA1_SGD (-2)."""
        if self.optimizer_type == "RMSprop":
            return -1.0, 0.99, 1e-8
        if self.optimizer_type == "SGD":
            return -2.0, 0.0, 0.0
        return self.betas[0], self.betas[1], self.eps

    def reset_state(self):
        """A fresh optimizer, like the deep_learning_setup() call of meta_weights_init('random') (trainer.py:356-359)."""
        self.exp_avg.zero_()
        self.exp_avg_sq.zero_()
        self.step = 0

    def check_status(self, value=None):
        """Raises MvnError if a training launch since the last check abandoned its device-wide barrier (its weights are NaN):
        the counterpart of the reference's NaN guard (trainer.py:496-498), which prints and skips the step.  Costs one
        4-byte device-to-host copy, and only when a kernel launch is outstanding: call it where the host synchronises anyway
        (harness.eval_by_word does after every block's `ser`).  value: the status word when the caller has just read it
        (sync_words[1], together with its own word) -- no copy then."""
        if self._unchecked and self.status is not None:
            self._unchecked = False
            if int(self.status.item() if value is None else value) != 0:
                self.status.zero_()
                raise _lib.MvnError("OnlineTrainer: " + _lib.load().mvn_strerror(-7).decode())

    @torch.no_grad()
    def optimizer_step(self, grads):
        """One step of the configured optimizer (torch.optim defaults, as deep_learning_setup builds them) from a list of
        gradients: Adam (shared with the kernels), RMSprop (alpha 0.99, eps 1e-8, square average kept in exp_avg_sq) or SGD."""
        if self.optimizer_type == "Adam":
            return self.adam_step(grads)
        self.step += 1
        off = 0
        for p, g in zip(self.params, grads):
            n = p.numel()
            if self.optimizer_type == "SGD":
                p.data.add_(g, alpha=-self.lr)
            else:
                v = self.exp_avg_sq[off:off + n].view_as(p)
                v.mul_(0.99).addcmul_(g, g, value=0.01)
                p.data.addcdiv_(g, v.sqrt().add_(1e-8), value=-self.lr)
            off += n

    @torch.no_grad()
    def adam_step(self, grads):
        """One torch.optim.Adam step (amsgrad off, no weight decay) on the detector from a list of gradients, using and
        updating the SAME exp_avg / exp_avg_sq / step as the online-training kernel (the reference has one optimizer for
        run_train_loop and meta_train_loop, trainer.py:163-173,452,503)."""
        self.step += 1
        b1, b2 = self.betas
        bc1, bc2 = 1.0 - b1 ** self.step, 1.0 - b2 ** self.step
        step_size = self.lr / bc1
        off = 0
        for p, g in zip(self.params, grads):
            n = p.numel()
            m, v = self.exp_avg[off:off + n].view_as(p), self.exp_avg_sq[off:off + n].view_as(p)
            m.mul_(b1).add_(g, alpha=1.0 - b1)
            v.mul_(b2).addcmul_(g, g, value=1.0 - b2)
            denom = (v.sqrt() / (bc2 ** 0.5)).add_(self.eps)
            p.data.addcdiv_(m, denom, value=-step_size)
            off += n

    @torch.no_grad()
    def adam_step_device(self, grads, step_t: torch.Tensor):
        """adam_step with the step counter and the bias corrections on the device (`step_t`: 0-d float64 tensor holding
        the number of steps taken so far), so that the whole update can sit inside a captured hipGraph
        (meta.GraphedMetaStep).  The caller keeps `self.step` in sync."""
        b1, b2 = self.betas
        step_t.add_(1.0)
        bc1 = 1.0 - torch.pow(b1, step_t)
        bc2 = 1.0 - torch.pow(b2, step_t)
        step_size = (self.lr / bc1).to(torch.float32)
        bc2_sqrt = torch.sqrt(bc2).to(torch.float32)
        off = 0
        for p, g in zip(self.params, grads):
            n = p.numel()
            m, v = self.exp_avg[off:off + n].view_as(p), self.exp_avg_sq[off:off + n].view_as(p)
            m.mul_(b1).add_(g, alpha=1.0 - b1)
            v.mul_(b2).addcmul_(g, g, value=1.0 - b2)
            denom = (v.sqrt() / bc2_sqrt).add_(self.eps)
            p.data.sub_((m / denom) * step_size)
            off += n

    def maml_training(self, rx_words: torch.Tensor, tx_words: torch.Tensor, support_idx: torch.Tensor,
                      query_idx: torch.Tensor, meta_lr: float, MAML: bool = True, return_loss: bool = False, labels: torch.Tensor = None):
        """n online meta-learning steps (trainer.py:425-453 = meta.meta_train_loop) in ONE kernel launch.
        rx_words / tx_words [Nw, T]: buffered received words and their (re-encoded or detected) transmitted words;
        support_idx [n, W], query_idx [n]: the words of every step (negative indices count from the end, like the
        reference's fancy indexing).  Uses and advances the same Adam state as online_training.
        labels: int32 [Nw, T], the trellis states of tx_words when the caller already has them (the block-step kernel writes them
        next to the word it buffers); None: calculate_states(tx_words)."""
        if self.optimizer_type != "Adam":  # (harness.eval_by_word runs the meta-learning updates of the other optimizers on autograd)
            raise NotImplementedError("the meta-learning kernel implements Adam; use meta.meta_train_loop for RMSprop / SGD")
        p = self.params
        dev = p[0].device
        _lib.require_gpu_tensor(rx_words, "rx_words")
        _lib.require_gpu_tensor(p[0], "detector parameters")
        rx = rx_words.detach().to(torch.float32).contiguous()
        Nw, T = rx.shape
        if labels is None:
            labels = calculate_states(self.memory_length, tx_words.detach().to(dev)).reshape(Nw, T).to(torch.int32).contiguous()
        elif labels.shape != (Nw, T) or labels.dtype != torch.int32 or not labels.is_contiguous():
            raise ValueError("labels: contiguous int32 [Nw, T]")
        sup = torch.remainder(support_idx.to(dev).reshape(query_idx.numel(), -1), Nw).to(torch.int32).contiguous()
        qry = torch.remainder(query_idx.to(dev).reshape(-1), Nw).to(torch.int32).contiguous()
        n, W = sup.shape
        for t in p:
            if not t.data.is_contiguous() or t.dtype != torch.float32:
                raise ValueError("ViterbiNet parameters must be contiguous fp32")
        loss = torch.empty(n, dtype=torch.float32, device=dev) if return_loss else None
        ws = self._workspace(p[5].numel(), dev)
        with torch.cuda.device(dev):
            rc = _lib.load().mvn_vnet_maml_train_ws_f32(_lib.ptr(rx), _lib.ptr(labels), T, _lib.ptr(sup), W, _lib.ptr(qry), n,
                                                        *[_lib.ptr(t.data) for t in p], _lib.ptr(self.exp_avg),
                                                        _lib.ptr(self.exp_avg_sq), self.step, meta_lr, 1 if MAML else 0,
                                                        self.lr, self.betas[0], self.betas[1], self.eps, _lib.ptr(loss),
                                                        p[5].numel(), _lib.ptr(ws), ws.numel(), _lib.ptr(self.status),
                                                        _lib.current_stream(dev))
        _lib.check(rc, "mvn_vnet_maml_train_ws_f32")
        self._unchecked = True
        self.step += n
        return loss

    def _workspace(self, S: int, dev) -> torch.Tensor:
        """Device scratch for the gradient exchange of the one-workgroup-per-chunk training kernels
        (mvn_vnet_train_workspace_bytes; about 3 MB), allocated once per trainer."""
        ws = getattr(self, "_ws", None)
        if ws is None or ws.device != dev:
            ws = self._ws = torch.empty(_lib.load().mvn_vnet_train_workspace_bytes(S), dtype=torch.uint8, device=dev)
        return ws

    def select_batches(self, T: int, iterations: int) -> torch.Tensor:
        """`iterations` minibatches drawn like select_batch (trainer.py:542): torch.multinomial with weights
        arange(T) (sample 0 is never drawn), without replacement, all iterations in one call."""
        dev = self.params[0].device
        w = torch.arange(T, dtype=torch.float32, device=dev).expand(iterations, T)
        return torch.multinomial(w, self.train_minibatch_size).to(torch.int32)

    def online_training(self, tx: torch.Tensor, rx: torch.Tensor, iterations: int = 200, batch_idx: torch.Tensor = None,
                        full_word: bool = False, return_loss: bool = False, labels: torch.Tensor = None):
        """tx [1,T] (re-encoded / detected word), rx [1,T] received word (vnet_trainer.py:49-60).
        full_word=True uses every sample each iteration (the Meta-ViterbiNet variant, metavnet_trainer.py:41-64).
        labels: int32 [T], the trellis states of tx when the caller already has them (kernel route only); None: calculate_states(tx)."""
        p = self.params
        dev = p[0].device
        # the one-launch kernel keeps parameters, gradient and a chunk's activations in LDS: n_states <= 128 (up to 32 with both
        # moments beside them; 64 / 128: the moments stay in global memory)
        if p[5].numel() > 128 or not self.use_kernel:
            return self._online_training_autograd(tx, rx, iterations, batch_idx, full_word, return_loss)
        _lib.require_gpu_tensor(rx, "rx")
        _lib.require_gpu_tensor(p[0], "detector parameters")
        y = rx.detach().to(torch.float32).reshape(-1).contiguous()
        T = y.numel()
        if labels is None:
            labels = calculate_states(self.memory_length, tx.detach().to(dev).reshape(1, -1)).to(torch.int32).contiguous()
        elif labels.numel() != T or labels.dtype != torch.int32 or not labels.is_contiguous():
            raise ValueError("labels: contiguous int32 [T]")
        if full_word:
            idx, M = None, 0
        else:
            idx = (self.select_batches(T, iterations) if batch_idx is None else batch_idx.to(device=dev, dtype=torch.int32)).contiguous()
            M = idx.shape[1]
            if idx.shape[0] != iterations:
                raise ValueError("batch_idx must be [iterations, M]")
        for t in p:
            if not t.data.is_contiguous() or t.dtype != torch.float32:
                raise ValueError("ViterbiNet parameters must be contiguous fp32")
        loss = torch.empty(iterations, dtype=torch.float32, device=dev) if return_loss else None
        S = p[5].numel()
        ws = self._workspace(S, dev)
        b1, b2, eps = self.kernel_optimizer_args()
        with torch.cuda.device(dev):
            rc = _lib.load().mvn_vnet_online_train_ws_f32(_lib.ptr(y), _lib.ptr(labels), T, _lib.ptr(idx), M, iterations,
                                                          *[_lib.ptr(t.data) for t in p], _lib.ptr(self.exp_avg),
                                                          _lib.ptr(self.exp_avg_sq), self.step, self.lr, b1, b2, eps,
                                                          _lib.ptr(loss), S, _lib.ptr(ws), ws.numel(),
                                                          _lib.ptr(self.status), _lib.current_stream(dev))
        _lib.check(rc, "mvn_vnet_online_train_ws_f32")
        self._unchecked = True
        self.step += iterations
        return loss

    def _online_training_autograd(self, tx, rx, iterations, batch_idx, full_word, return_loss):
        """The same loop on stock PyTorch autograd (run_train_loop, trainer.py:492-505: forward 'train', CrossEntropy over the
        selected samples, backward, Adam on the shared exp_avg / exp_avg_sq / step): the route for memory_length 8 (256 states), whose
        parameter set does not fit the training kernel's LDS image, and for use_kernel=False.  Same draws as the kernel path (select_batches)."""
        import torch.nn.functional as F

        p = self.params
        dev = p[0].device
        S = p[5].numel()
        y = rx.detach().to(device=dev, dtype=torch.float32).reshape(1, -1)
        T = y.shape[1]
        labels = calculate_states(self.memory_length, tx.detach().to(dev).reshape(1, -1)).reshape(-1).long()
        idx = None
        if not full_word:
            idx = (self.select_batches(T, iterations) if batch_idx is None else batch_idx.to(dev)).long()
            if idx.shape[0] != iterations:
                raise ValueError("batch_idx must be [iterations, M]")
        losses = []
        for it in range(iterations):
            logits = self.detector(y, "train").reshape(-1, S)
            loss = F.cross_entropy(logits, labels) if full_word else F.cross_entropy(logits[idx[it]], labels[idx[it]])
            self.optimizer_step(torch.autograd.grad(loss, p))
            if return_loss:
                losses.append(loss.detach())
        return torch.stack(losses).to(torch.float32) if return_loss else None
