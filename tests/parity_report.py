"""Every oracle comparison of the -m gpu tests leaves its MEASURED distance here, next to the bar it was held to
("state the parity slack as measured + margin, and show it").

    from parity_report import check_u8, check_f32

    check_u8("tiled 2x 70x75 t32", got, want, vs="fp32 oracle", max_lsb=2, min_psnr=50)

records max |diff| (LSB), PSNR (dB) and the share of differing samples, THEN asserts the bars.  conftest.py writes the
records to gpurun_out/parity_report.json at the end of the session and prints one line per record in pytest's terminal
summary (so that the driver's `-q` log shows them); tools/parity_slack.py turns a report into tests/golden/parity_slack.json,
the committed measurements the product-mode bars are derived from (bar = measured maximum + margin, per model and route).
"""
import json
import os

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
RECORDS = []
_SLACK = None


def psnr_u8(a, b):
    d = a.astype(np.float64) - b.astype(np.float64)
    mse = float((d * d).mean()) if d.size else 0.0
    return 99.0 if mse == 0 else float(10 * np.log10(255.0 ** 2 / mse))


def slack(model, route, what, fallback):
    """Bar for `what` ("u8_differ_share", "layer_rel", "f32_abs") of `model` on `route` from the committed sweep:
    measured maximum + margin (tests/golden/parity_slack.json, written by tools/parity_slack.py from a GPU run's report);
    `fallback` where the file has no entry (a new model / route: the round number it replaces)."""
    global _SLACK
    if _SLACK is None:
        try:
            _SLACK = json.load(open(os.path.join(ROOT, "tests", "golden", "parity_slack.json")))
        except (OSError, ValueError):
            _SLACK = {}
    e = _SLACK.get("bars", {}).get(f"{model}/{route}/{what}")
    return float(e["bar"]) if e else fallback


def record(name, **fields):
    rec = {"name": name}
    rec.update({k: (round(v, 6) if isinstance(v, float) else v) for k, v in fields.items()})
    RECORDS.append(rec)
    return rec


def check_u8(name, got, want, vs, max_lsb, min_psnr=None, max_share=None, model=None, route=None):
    """u8 frames: record, then hold to the bars (max |diff| in LSB, PSNR in dB, share of samples that differ)."""
    assert got.shape == want.shape and got.dtype == np.uint8, (name, got.shape, want.shape, got.dtype)
    d = np.abs(got.astype(np.int16) - want.astype(np.int16))
    worst = int(d.max()) if d.size else 0
    share = float((d > 0).mean()) if d.size else 0.0
    p = psnr_u8(got, want)
    record(name, kind="u8", vs=vs, model=model, route=route, samples=int(d.size), max_lsb=worst, psnr_db=p, differ_share=share,
           bar_max_lsb=max_lsb, bar_min_psnr_db=min_psnr, bar_max_share=max_share)
    assert worst <= max_lsb, (name, vs, "max |diff|", worst, "bar", max_lsb)
    if min_psnr is not None:
        assert p >= min_psnr, (name, vs, "PSNR", p, "bar", min_psnr)
    if max_share is not None:
        assert share <= max_share, (name, vs, "differing share", share, "bar", max_share)
    return worst, p, share


def check_f32(name, got, want, vs, max_abs, model=None, route=None, what="f32_abs", scale=None):
    """float blobs: record max |diff| (and, with `scale`, relative to the blob's range), then hold to max_abs"""
    err = float(np.abs(got - want).max()) if got.size else 0.0
    record(name, kind="f32", vs=vs, model=model, route=route, what=what, max_abs_err=err,
           rel_to_range=(err / scale if scale else None), bar_max_abs=float(max_abs))
    assert err <= max_abs, (name, vs, "max |err|", err, "bar", max_abs)
    return err


def write(path=None):
    if not RECORDS:
        return None
    out_dir = os.path.join(ROOT, "gpurun_out")
    path = path or os.path.join(out_dir, "parity_report.json")
    try:
        os.makedirs(os.path.dirname(path), exist_ok=True)
        with open(path, "w") as f:
            json.dump({"records": RECORDS}, f, indent=1)
    except OSError:
        return None
    return path


def summary_lines():
    for r in RECORDS:
        if r.get("kind") == "u8":
            yield ("parity %-58s vs %-34s max %d LSB (bar %s)  PSNR %6.2f dB (bar %s)  differ %.3f %% (bar %s)" % (
                r["name"][:58], r["vs"][:34], r["max_lsb"], r["bar_max_lsb"], r["psnr_db"], r["bar_min_psnr_db"],
                100 * r["differ_share"], ("%.1f %%" % (100 * r["bar_max_share"])) if r["bar_max_share"] is not None else "-"))
        else:
            yield ("parity %-58s vs %-34s max |err| %.3e (bar %.3e)%s" % (
                r["name"][:58], r["vs"][:34], r["max_abs_err"], r["bar_max_abs"],
                ("  = %.2e of the range" % r["rel_to_range"]) if r.get("rel_to_range") else ""))
