"""CPU ORACLE -- TEST INFRASTRUCTURE ONLY.

Independent numpy restatement of the CTC negative log-likelihood and its gradient w.r.t.
the log-probabilities (Graves et al. 2006 alpha/beta recursion, log space) with the
conventions of ``torch.nn.functional.ctc_loss`` used by the build-defined step
(SURVEY.md section 3.3 / Appendix C): blank = 0, reduction 'mean' = per-sample NLL divided by
its target length then averaged over the batch, infeasible alignment -> 0 (zero_infinity).
The reference itself contains no CTC loss ("parity unpinned" inside the reference): this
file cross-checks PyTorch's CPU kernel, which is what the golden fixtures were made with.
"""
import numpy as np

NEG = -1e30


def _lse(a, b):
    m = np.maximum(a, b)
    return m + np.log(np.exp(a - m) + np.exp(b - m))


def ctc_nll_and_grad(log_probs, target):
    """log_probs [T,C] (already log-softmaxed), target [L] ints in 1..C-1.
    Returns (nll, d nll / d log_probs [T,C])."""
    T, C = log_probs.shape
    L = len(target)
    S = 2 * L + 1
    ext = np.zeros(S, dtype=np.int64)
    ext[1::2] = target
    alpha = np.full((T, S), NEG)
    beta = np.full((T, S), NEG)
    alpha[0, 0] = log_probs[0, 0]
    if S > 1:
        alpha[0, 1] = log_probs[0, ext[1]]
    for t in range(1, T):
        for s in range(S):
            a = alpha[t - 1, s]
            if s >= 1:
                a = _lse(a, alpha[t - 1, s - 1])
            if s >= 2 and ext[s] != 0 and ext[s] != ext[s - 2]:
                a = _lse(a, alpha[t - 1, s - 2])
            alpha[t, s] = a + log_probs[t, ext[s]]
    ll = alpha[T - 1, S - 1]
    if S > 1:
        ll = _lse(ll, alpha[T - 1, S - 2])
    beta[T - 1, S - 1] = log_probs[T - 1, 0]
    if S > 1:
        beta[T - 1, S - 2] = log_probs[T - 1, ext[S - 2]]
    for t in range(T - 2, -1, -1):
        for s in range(S):
            b = beta[t + 1, s]
            if s + 1 < S:
                b = _lse(b, beta[t + 1, s + 1])
            if s + 2 < S and ext[s + 2] != 0 and ext[s + 2] != ext[s]:
                b = _lse(b, beta[t + 1, s + 2])
            beta[t, s] = b + log_probs[t, ext[s]]
    grad = np.zeros((T, C))
    if ll < -1e29:                      # infeasible -> zero_infinity
        return 0.0, grad
    for t in range(T):
        acc = np.full(C, NEG)
        for s in range(S):
            acc[ext[s]] = _lse(acc[ext[s]], alpha[t, s] + beta[t, s])
        grad[t] = -np.exp(acc - ll - log_probs[t])
    return -ll, grad


def ctc_mean(log_probs_tbc, targets_flat, target_lengths):
    """Batch 'mean' reduction; returns (loss, grad [T,B,C])."""
    T, B, C = log_probs_tbc.shape
    off = 0
    total = 0.0
    grad = np.zeros_like(log_probs_tbc, dtype=np.float64)
    for b in range(B):
        L = int(target_lengths[b])
        nll, g = ctc_nll_and_grad(log_probs_tbc[:, b].astype(np.float64),
                                  np.asarray(targets_flat[off:off + L]))
        off += L
        total += nll / max(L, 1) / B
        grad[:, b] = g / max(L, 1) / B
    return total, grad
