"""Observed parity margins: the step-level parity tests record, per tensor, how far the HIP path actually lands from
the oracle -- max |got - ref| relative to the tensor's scale max |ref| (and, for weights after Adam steps, in units of the learning rate:
Adam moves a weight by ~lr per step whatever the gradient's size, so lr is the natural scale of an error there) -- into
gpurun_out/parity_margins.json on the box the tests run on; the builder copies that file to profiles/rNN_parity_margins.json and sets each
test's tolerance to <= 2x the observed value.  Test infrastructure only."""
import atexit
import json
import os
from pathlib import Path

import numpy as np

_ROOT = Path(os.environ.get("GRAFT_REPO_ROOT", Path(__file__).resolve().parent.parent))
_OUT = _ROOT / "gpurun_out" / "parity_margins.json"
_seen = {}


def record(test: str, tensor: str, got, ref, lr: float | None = None) -> dict:
    got, ref = np.asarray(got, np.float64), np.asarray(ref, np.float64).reshape(np.shape(got))
    diff = float(np.abs(got - ref).max()) if got.size else 0.0
    scale = float(np.abs(ref).max()) if ref.size else 0.0
    nz = np.abs(ref) > 1e-3 * max(scale, 1e-30)
    rec = dict(max_abs_err=diff, ref_scale=scale, err_over_scale=(diff / scale if scale > 0 else 0.0),
               max_rel_err_on_elements_above_0p001_of_scale=float((np.abs(got - ref)[nz] / np.abs(ref)[nz]).max()) if nz.any() else 0.0)
    if lr is not None:
        rec["err_over_lr"] = diff / lr
    _seen.setdefault(test, {})[tensor] = rec
    return rec


def check(tensor: str, got, ref, bound: float | None, lr: float | None = None, test: str | None = None) -> dict:
    """Record the margin and hold it to `bound`: max |got - ref| <= bound * max |ref| -- or <= bound * lr when `lr` is given (weights
    after Adam steps).  The record keeps the WORST case over a test's parametrisations (key = the test function's name), so a bound of
    2x the recorded value covers all of them.  bound None: record only."""
    if test is None:
        test = os.environ.get("PYTEST_CURRENT_TEST", "unknown").split(" ")[0].split("::")[-1].split("[")[0]
    prev = _seen.get(test, {}).get(tensor)
    rec = record(test, tensor, got, ref, lr)
    key = "err_over_lr" if lr is not None else "err_over_scale"
    val = rec[key]
    if prev is not None and prev.get(key, 0.0) > val:
        _seen[test][tensor] = prev
    if bound is not None:
        assert val <= bound, f"{test}/{tensor}: {key} = {val:.3e} > bound {bound:.3e} ({rec})"
    return rec


def check_adam_weights(tensor: str, got, ref, v_ref, lr: float, steps: int, bound: float, test: str | None = None, v_floor: float = 1e-5) -> None:
    """Weights after `steps` Adam steps (optim.rs:99-110) against the oracle's.  A weight moves by lr m / (sqrt(v) + eps) per step: for
    an element whose gradient is a cancellation down to ~eps that quotient is O(1) whatever another summation order does to the gradient's
    last bits, so the elements whose gradients stand clear of eps (sqrt(v_ref) > v_floor) are held to `bound` x lr, and every element to
    2 lr per step (a sign flip of such a gradient) -- both recorded."""
    got, ref = np.asarray(got, np.float64).reshape(-1), np.asarray(ref, np.float64).reshape(-1)
    clear = np.sqrt(np.abs(np.asarray(v_ref, np.float64).reshape(-1))) > v_floor
    assert clear.any(), f"{tensor}: no element has a gradient clear of eps"
    check(f"{tensor}_clear_of_eps", got[clear], ref[clear], bound, lr=lr, test=test)
    check(tensor, got, ref, 2.0 * steps, lr=lr, test=test)


@atexit.register
def _flush():
    if not _seen:
        return
    try:
        _OUT.parent.mkdir(parents=True, exist_ok=True)
        old = json.loads(_OUT.read_text()) if _OUT.exists() else {}
        for k, v in _seen.items():
            old.setdefault(k, {}).update(v)   # (one pytest process per run: a re-run replaces its tests' records)
        _OUT.write_text(json.dumps(old, indent=1, sort_keys=True))
    except OSError:
        pass
