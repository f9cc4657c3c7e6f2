"""Batched SPARC smoothness (``mpinets/third_party/sparc.py``, used by ``Evaluator.calculate_smoothness``,
``mpinets/metrics.py:387-409``) for ragged batches of speed profiles, on whatever device the profiles live on.

The reference scores ONE profile per call with numpy; an evaluation run calls it twice per trajectory (joint space and
end-effector space).  Here a whole batch is scored at once: the profiles are grouped by FFT length (a profile of n samples
is zero-padded to ``2^(ceil(log2 n) + padlevel)`` bins: at most eight distinct lengths for trajectories of up to 150
waypoints), every group is one batched float64 FFT, and the frequency cut-off, the amplitude window and the arc length are
masks and sums over the bins.  Same arithmetic per profile as the reference (full frequency axis ``k * fs / nfft`` up to
fs -- at run_inference's 1 / 0.12 s the cut-off fc = 10 Hz lies above fs, so the mirrored half of the spectrum is part
of the curve, as it is there).  Pinned: ``tests/golden/sparc_golden.npz`` (the reference function's own outputs).
"""
from __future__ import annotations

from typing import Optional, Tuple

import torch


def fft_length(n: int, padlevel: int = 4) -> int:
    """``int(pow(2, ceil(log2(n)) + padlevel))`` (sparc.py:97) in integers: the next power of two >= n, times 2^padlevel."""
    assert n >= 1
    return (1 << (n - 1).bit_length()) << padlevel


@torch.no_grad()
def sparc_batched(movement: torch.Tensor, lengths: Optional[torch.Tensor], fs: float, padlevel: int = 4, fc: float = 10.0,
                  amp_th: float = 0.05) -> torch.Tensor:
    """:param movement: [B, T] speed profiles (any float dtype; evaluated in float64); samples at t >= lengths[b] are ignored
    :param lengths: [B] number of valid samples per profile (>= 1), or None for T everywhere
    :param fs: sampling rate of the profiles
    :returns: [B] float64 spectral arc lengths (<= 0; 0 for a profile that is zero everywhere, like the reference;
        NaN where no bin below the cut-off reaches the amplitude threshold -- the reference raises there)
    """
    assert movement.ndim == 2
    B, T = movement.shape
    dev = movement.device
    m = movement.to(torch.float64)
    ln = torch.full((B,), T, dtype=torch.int64, device=dev) if lengths is None else lengths.to(device=dev, dtype=torch.int64)
    assert bool((ln >= 1).all()) and bool((ln <= T).all()), "lengths must be in [1, T]"
    m = torch.where(torch.arange(T, device=dev)[None, :] < ln[:, None], m, torch.zeros((), dtype=torch.float64, device=dev))
    out = torch.zeros(B, dtype=torch.float64, device=dev)
    # np.allclose(movement, 0) with its defaults: every |x| <= 1e-8 (sparc.py:93-95)
    moving = (m.abs() > 1e-8).any(dim=1)
    ln_host = ln.cpu().tolist()
    by_nfft = {}
    for b, n in enumerate(ln_host):
        by_nfft.setdefault(fft_length(int(n), padlevel), []).append(b)
    for nfft, rows in by_nfft.items():
        idx = torch.as_tensor(rows, dtype=torch.int64, device=dev)
        mag = torch.fft.fft(m[idx], n=nfft, dim=1).abs()  # (zero-pads; a longer row is zero past its own length anyway)
        mag = mag / mag.max(dim=1, keepdim=True).values.clamp_min(torch.finfo(torch.float64).tiny)
        k = torch.arange(nfft, device=dev)
        freq = k.to(torch.float64) * (fs / nfft)  # np.arange(0, fs, fs / nfft): start + k * step
        loud = (mag >= amp_th) & (freq <= fc)[None, :]
        has = loud.any(dim=1)
        first = loud.to(torch.int8).argmax(dim=1)
        last = nfft - 1 - loud.flip(1).to(torch.int8).argmax(dim=1)
        span = (freq[last] - freq[first]).clamp_min(torch.finfo(torch.float64).tiny)  # (first == last: no segment at all)
        seg = (k[None, :-1] >= first[:, None]) & (k[None, :-1] < last[:, None])
        df = (freq[1:] - freq[:-1])[None, :] / span[:, None]
        dm = mag[:, 1:] - mag[:, :-1]
        arc = torch.where(seg, torch.sqrt(df * df + dm * dm), torch.zeros((), dtype=torch.float64, device=dev)).sum(dim=1)
        res = torch.where(has, -arc, torch.full_like(arc, float("nan")))
        out[idx] = torch.where(moving[idx], res, torch.zeros_like(res))
    return out


@torch.no_grad()
def speed_profile(x: torch.Tensor, dt: float) -> torch.Tensor:
    """``np.linalg.norm(np.diff(x, 1, axis=0) / dt, axis=1)`` (metrics.py:397, 405) for a batch [B, T, D] -> [B, T - 1]."""
    x = x.to(torch.float64)
    return torch.linalg.vector_norm((x[:, 1:] - x[:, :-1]) / dt, dim=2)


@torch.no_grad()
def trajectory_smoothness(trajectories: torch.Tensor, eff_positions: torch.Tensor, lengths: Optional[torch.Tensor],
                          dt: float) -> Tuple[torch.Tensor, torch.Tensor]:
    """``Evaluator.calculate_smoothness`` for a batch: SPARC of the joint-space speed profile of ``trajectories``
    [B, T, 7] and of the end-effector positions' [B, T, 3]; ``lengths`` [B] valid waypoints (>= 2).
    -> (config_sparc [B], eff_sparc [B]) float64; "smooth" in the reference's summary means both < -1.6 (metrics.py:589-594)."""
    B, T, _ = trajectories.shape
    assert T >= 2, "a trajectory needs two waypoints to have a speed"
    n = None if lengths is None else (lengths.to(torch.int64) - 1).clamp(min=1)
    fs = 1.0 / dt
    return (sparc_batched(speed_profile(trajectories, dt), n, fs), sparc_batched(speed_profile(eff_positions, dt), n, fs))
