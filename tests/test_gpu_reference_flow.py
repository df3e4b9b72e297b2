"""The reference's own `eval_by_word` WITH its update branches, replayed draw for draw (golden G12).

tests/golden/make_golden.py:g12_by_word_with_updates ran the unmodified reference (trainer.py:267-354) over 75 blocks from the
reference-trained weights of G7, once as VNETTrainer (self-supervised minibatch training, vnet_trainer.py:49-60) and once as
METAVNETTrainer (online meta-learning every 5 blocks + whole-word training from the saved weights, metavnet_trainer.py:52-64),
recording every torch.multinomial / torch.randint draw it made.  Here the same words go through mvn.eval_by_word with a draws
object that hands those recorded values back in call order, on the HIP kernels and on the torch-autograd path of the same host
code.  What must come out: the reference's ser_by_word, block for block (the ser decides which blocks are buffered and trained
on, so one differing block would also misalign the recorded draws — the draws object raises when that happens), every recorded
draw consumed, and the final weights within the training tolerance of test_gpu_replay.py."""
import numpy as np
import pytest
import torch

import meta_viterbinet_amd as mvn

pytestmark = pytest.mark.gpu


class RecordedDraws:
    """trials.TrialDraws' interface over the draws the reference made (in its call order)."""

    def __init__(self, multinomial, randint_high, randint, device):
        self.multinomial, self.randint_high, self.randint, self.device = multinomial, randint_high, randint, device
        self.m_at = self.r_at = 0

    def batches(self, count, n_blocks, T, iterations, M):
        rows = self.multinomial[self.m_at:self.m_at + iterations]
        if rows.shape != (iterations, M):
            raise AssertionError(f"block {count} trains, the reference's recorded minibatches are used up ({self.m_at} of {len(self.multinomial)})")
        assert rows.max() < T
        self.m_at += iterations
        return torch.as_tensor(rows, dtype=torch.int32, device=self.device)

    def j_hat(self, high, size):
        return self.j_hat_update(high, 1, size)

    def j_hat_update(self, high, iterations, size):
        out = []
        for _ in range(iterations):
            if self.r_at >= len(self.randint):
                raise AssertionError("a meta-learning update the reference did not make")
            assert self.randint_high[self.r_at] == high, (self.r_at, int(self.randint_high[self.r_at]), high)  # same buffer length
            assert self.randint.shape[1] == size
            out.append(np.unique(self.randint[self.r_at]))  # torch.unique(torch.randint(...)) (trainer.py:337)
            self.r_at += 1
        return np.concatenate(out)

    def used_up(self):
        return self.m_at == len(self.multinomial) and self.r_at == len(self.randint)


def _run(g, g7, tag, dev, hip):
    tx = torch.tensor(g[f"{tag}_tx"], device=dev).float()
    rx = torch.tensor(g[f"{tag}_rx"], device=dev)
    ss_it, meta_it, j_num, meta_sub, _, subframes, nsym = [int(v) for v in g[f"{tag}_meta"]]
    T = rx.shape[1]
    det = mvn.VNETDetector(16, {"train": T, "val": T}).to(dev)
    with torch.no_grad():
        for p, i in zip(det.parameters(), range(6)):
            p.copy_(torch.as_tensor(g7[f"w{i}"]))
    tr = mvn.OnlineTrainer(det, 4, use_kernel=hip)
    draws = RecordedDraws(g[f"{tag}_multinomial"], g[f"{tag}_randint_high"], g[f"{tag}_randint"], dev)
    kw = dict(self_supervised=True, online_trainer=tr, self_supervised_iterations=ss_it, ser_thresh=0.02, draws=draws,
              hip_meta=hip, graphed_meta=False)
    if tag == "meta":
        kw.update(online_meta=True, meta_detector=mvn.META_VNETDetector(16, {"train": T, "val": T}), meta_lr=0.1, MAML=True,
                  window_size=1, meta_train_iterations=meta_it, meta_j_num=j_num, meta_subframes=meta_sub,
                  meta_style_online_training=True)
    ser = mvn.eval_by_word(det, tx, rx, 9.0, 0.2, nsym, subframes, **kw)
    return ser, draws, [p.detach().cpu().numpy() for p in det.parameters()]


@pytest.mark.parametrize("tag", ["selfsup", "meta"])
@pytest.mark.parametrize("hip", [True, False], ids=["hip_kernels", "torch_autograd"])
def test_reference_by_word_flow_with_updates(golden, dev, tag, hip):
    g, g7 = golden("g12_by_word_with_updates"), golden("g7_by_word")
    ser, draws, w = _run(g, g7, tag, dev, hip)
    ref = g[f"{tag}_ser_by_word"]
    assert ser.shape == ref.shape == (75,)
    assert np.array_equal(ser, ref), (np.flatnonzero(ser != ref), ser[ser != ref], ref[ser != ref])
    assert draws.used_up(), (draws.m_at, len(draws.multinomial), draws.r_at, len(draws.randint))
    # the weights the reference ended with: 59 x 12 minibatch Adam steps / 14 meta-learning updates + whole-word training later,
    # its CPU kernels against these (different summation orders from the first matmul on): the per-step tolerance of
    # test_gpu_replay.py does not apply to a whole run; what holds is closeness at the scale of the update itself
    moved = max(float(np.abs(g[f"{tag}_w1_{i}"] - g7[f"w{i}"]).max()) for i in range(6))
    worst = max(float(np.abs(w[i] - g[f"{tag}_w1_{i}"]).max()) for i in range(6))
    print(f"g12 {tag} {'hip' if hip else 'torch'}: ser identical on 75 blocks; weights moved {moved:.4f} from the start, "
          f"end {worst:.2e} from the reference's")
    assert moved > 1e-3 and worst <= 0.02 * moved
