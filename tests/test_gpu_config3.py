"""BASELINE config 3 ("ModelNet10 training, surfaceNetUpdatedEdgeFilters, bf16") as a functional check: bf16 STORAGE trains like fp32 storage.

The per-step gradients of the bf16-storage path are pinned to a rounding model (tests/test_gpu_scale.py) whose distance from the fp64 oracle -- the
price of the format -- is 5-9 % rms at the first conv layers.  What that price means for TRAINING is measured here: the
Updated model at the reference's ModelNet10 widths and batch size (configs/modelnet.yaml:44,56; model learning/surfaceNetUpdatedEdgeFilters.py:216-251;
step learning/runModel.py:264-282), 300 Adam steps (lr 0.005, runModel.py:95-99) from the same initial weights on the same blocks, once per storage type."""
import os
import sys

import pytest

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))

pytestmark = pytest.mark.gpu

# The stated band.  The loss falls 300 x over the run, 4-5 x per 25 steps during the first 100: there a lag of three steps is a
# 20 % gap between the 25-step means (measured 20.5 % at steps 25-50, bf16 behind), so the band is BAND_EARLY for steps < 100 and BAND_LATE from step
# 100 on (measured: <= 5.8 %, 0.5-1.6 % at the end); both runs must reduce the loss by at least REDUCTION (measured: 300 x).
BAND_EARLY = 0.30
BAND_LATE = 0.10
REDUCTION = 0.05


@pytest.mark.parametrize("widths,batch", [((128, 256, 512, 1024), 1024), ((64, 128, 128, 128), 2048)])
def test_bf16_storage_loss_curve_tracks_fp32_storage(widths, batch):
    from config3_convergence import run
    r = run(widths=widths, batch=batch, steps=300, points=10000 if batch == 1024 else 20000, window=25)
    print("config 3 convergence, widths %s batch %d: fp32 %s" % (list(widths), batch, r["loss_f32"]))
    print("                                              bf16 %s" % (r["loss_bf16"],))
    print("   relative gap of the 25-step means: max %.4f before step 100, %.4f after (last %.4f); loss %.5f -> fp32 %.5f / bf16 %.5f"
          % (r["max_rel_gap_early"], r["max_rel_gap_late"], r["last_rel_gap"], r["loss_f32_first"], r["loss_f32_last"], r["loss_bf16_last"]))
    try:      # scratch copy for BASELINE.md's table
        import json
        root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
        os.makedirs(os.path.join(root, "gpurun_out"), exist_ok=True)
        with open(os.path.join(root, "gpurun_out", "config3_convergence_%s.json" % "_".join(map(str, widths))), "w") as f:
            json.dump(r, f)
    except OSError:
        pass
    assert r["finite"]
    assert r["loss_f32_last"] <= REDUCTION * r["loss_f32_first"] and r["loss_bf16_last"] <= REDUCTION * r["loss_f32_first"], r
    assert r["max_rel_gap_early"] <= BAND_EARLY and r["max_rel_gap_late"] <= BAND_LATE, r
