"""GPU: inference path (Test_KITTI.py forward + ms_pp) against the golden recorded from the reference, the
reference-format checkpoint round trip, and the entry-point scripts in --synthetic mode."""
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from fal_net_amd import inference, synthetic  # noqa: E402
from fal_net_amd import myUtils as utils  # noqa: E402
from fal_net_amd.models import FAL_netB  # noqa: E402

DEV = "cuda"
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))


def rel(a, b):
    a, b = torch.as_tensor(a).detach().double().cpu(), torch.as_tensor(b).detach().double().cpu()
    return float((a - b).abs().max() / b.abs().max().clamp_min(1e-12))


def test_ms_pp_vs_golden(golden_dir):
    g = np.load(os.path.join(golden_dir, "g5_ms_pp.npz"))
    left, right, mn, mx = synthetic.synthetic_pair(1, 96, 320, seed=int(g["seed"]))
    m = FAL_netB({"state_dict": synthetic.seeded_falnetb_state_dict(49)}, 49).to(DEV).eval()
    with torch.no_grad():
        disp = m(left.to(DEV), mn.to(DEV), mx.to(DEV))
        pp = inference.ms_pp(left.to(DEV), m, disp, mn.to(DEV), mx.to(DEV))
    assert rel(disp, g["disp"]) < 1e-4
    assert rel(pp, g["ms_pp"]) < 2e-4  # the reference flips through grid_sample (leaks ~1e-6 bilinear weights)


def test_checkpoint_roundtrip_reference_format(tmp_path):
    m = FAL_netB({"state_dict": synthetic.seeded_falnetb_state_dict(7)}, 7).to(DEV).eval()
    left, right, mn, mx = synthetic.synthetic_pair(1, 64, 128, seed=3)
    with torch.no_grad():
        d0 = m(left.to(DEV), mn.to(DEV), mx.to(DEV))
    utils.save_checkpoint({"epoch": 1, "m_model": "FAL_netB", "state_dict": m.state_dict(), "best_rmse": -1}, True, str(tmp_path))
    data = torch.load(os.path.join(tmp_path, "model_best.pth.tar"), map_location="cpu")
    assert list(data["state_dict"].keys()) == list(synthetic.falnetb_param_shapes(7).keys())
    import models
    m2 = models.__dict__[data["m_model"]](data, no_levels=7).to(DEV).eval()
    with torch.no_grad():
        d1 = m2(left.to(DEV), mn.to(DEV), mx.to(DEV))
    # two module instances autotune their launches independently (different summation orders): equal to rounding
    assert rel(d1, d0) < 1e-5


@pytest.mark.parametrize("cmd", [
    ["Test_KITTI.py", "--height", "96", "--width", "320", "--iters", "2", "--dtype", "f32"],
    ["Train_Stage1_K.py", "--synthetic", "--epochs", "1", "--epoch_size", "2", "-b", "1", "-ch", "64", "-cw", "128", "-p", "1"],
    ["Train_Stage1_K.py", "--synthetic", "--gpu-augment", "--epochs", "1", "--epoch_size", "2", "-b", "2", "-ch", "64", "-cw", "128", "-p", "1"],
    ["Train_Stage1_Kslow.py", "--synthetic", "--epochs", "1", "--epoch_size", "2", "-b", "1", "-ch", "64", "-cw", "128", "-p", "1"],
    ["Train_Stage2_K.py", "--synthetic", "--epochs", "1", "--epoch_size", "2", "-b", "1", "-ch", "64", "-cw", "128", "-p", "1", "-no_levels", "7"],
    # the other two model variants through the same scripts (FAL_netA: 3x1 / 1x3 residual convs + its align_corners=False right mask in Stage 2)
    ["Test_KITTI.py", "-m", "FAL_netA", "-no_levels", "33", "--height", "96", "--width", "320", "--iters", "2", "--dtype", "f32"],
    ["Train_Stage1_K.py", "-mm", "FAL_netC", "-no_levels", "33", "--synthetic", "--epochs", "1", "--epoch_size", "2", "-b", "1", "-ch", "64", "-cw", "128", "-p", "1"],
    ["Train_Stage2_K.py", "-mm", "FAL_netA", "--synthetic", "--epochs", "1", "--epoch_size", "2", "-b", "1", "-ch", "64", "-cw", "128", "-p", "1", "-no_levels", "7"],
])
def test_entry_scripts_synthetic(cmd, tmp_path):
    extra = ["--save-path", str(tmp_path)] if cmd[0].startswith("Train_Stage1_K") else []
    r = subprocess.run([sys.executable, os.path.join(ROOT, cmd[0])] + cmd[1:] + extra, capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    assert "{" in r.stdout


def test_kitti_native_size_forward_and_ms_pp_vs_oracle():
    """Test_KITTI.py's default input, 375 x 1242 (B = 1, f32): odd sizes at every level -- 188 -> 94 -> 47 -> 24 -> 12 -> 6, so
    every decoder stage resamples with a non-integer nearest ratio (FAL_netB.py:58) -- and the ms_pp second forward at
    250 x 828.  Forward disparity and the post-processed map against the CPU oracle (pinned by goldens G5 / G7 at small sizes)."""
    from oracle import falnet_oracle as O
    left, right, mn, mx = synthetic.synthetic_pair(1, 375, 1242, seed=77)
    sd = synthetic.seeded_falnetb_state_dict(49)
    m = FAL_netB({"state_dict": sd}, 49).to(DEV).eval()
    with torch.no_grad():
        disp = m(left.to(DEV), mn.to(DEV), mx.to(DEV))
        pp = inference.ms_pp(left.to(DEV), m, disp, mn.to(DEV), mx.to(DEV))
        ref = O.falnet_forward(sd, left, mn, mx)
        ref_pp = O.ms_pp(sd, left, ref, mn, mx)
    assert rel(disp, ref) < 1e-4
    assert rel(pp, ref_pp) < 2e-4


@pytest.mark.parametrize("script,extra", [("Train_Stage1_K.py", []), ("Train_Stage2_K.py", ["-no_levels", "7", "--allow-seeded-teacher"])])
def test_entry_scripts_train_and_validate_on_generated_pngs(script, extra, tmp_path):
    """The real-data path end to end without KITTI: generated PNG tree + pair list -> loader workers decode -> uint8 upload ->
    GPU augmentation -> two optimiser steps -> validate() on two 375 x 1242 KITTI-2015-shaped pairs (RMSE, EPE, KITTI errors) ->
    checkpoint + model_best (Train_Stage1_K.py:137-160, :190-207, :279-347)."""
    import json
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from test_host_logic import _write_png_fixture
    root, lst = _write_png_fixture(tmp_path)
    save = tmp_path / "out"
    cmd = [sys.executable, os.path.join(ROOT, script), "-d", str(root), "--train_list", str(lst), "--epochs", "1", "-b", "2", "-ch", "64", "-cw", "128",
           "-p", "1", "-w", "2", "--dtype", "f32", "--save-path", str(save)] + extra
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-3000:]
    recs = [json.loads(l) for l in r.stdout.splitlines() if l.startswith("{")]
    steps = [x for x in recs if "iter" in x]
    val = [x for x in recs if "val_rmse" in x]
    assert len(steps) == 2 and all(np.isfinite(x["loss"]) for x in steps)  # 4 pairs / batch 2
    assert len(val) == 1 and 0 < val[0]["val_rmse"] < 255 and np.isfinite(val[0]["val_epe"]) and set(val[0]["kitti"]) == set(utils.kitti_error_names)
    assert os.path.isfile(save / "checkpoint.pth.tar") and os.path.isfile(save / "model_best.pth.tar")
    assert "=> 4 training pairs, 2 validation pairs" in r.stdout
    if script == "Train_Stage2_K.py":  # a real-data Stage-2 run without Stage-1 checkpoints must refuse (the reference torch.loads them)
        bad = subprocess.run([c for c in cmd if c != "--allow-seeded-teacher"], capture_output=True, text=True, timeout=300, cwd=ROOT)
        assert bad.returncode != 0 and "--fix_model" in (bad.stderr + bad.stdout)


def test_test_kitti_reference_command_line(tmp_path):
    """A reference-style invocation (reference Test_KITTI.py:36-60,119-122): `-m` is the model NAME, the checkpoint is composed as
    <-dt>/<-ts>/<-m><-dtl>, the model class comes from the checkpoint's m_model entry, and the results land in
    Test_Results/<tdataName>/<model>/<time_stamp>mspp relative to the working directory."""
    import json
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from test_host_logic import _write_png_fixture
    root, _ = _write_png_fixture(tmp_path, n_train=1, n_val=1)
    ckdir = tmp_path / "Kitti_stage2" / "10-18-15_42" / "FAL_netB,e20es,b4,lr5e-05"
    os.makedirs(ckdir)
    sd = synthetic.seeded_state_dict("A", 33)  # the checkpoint says FAL_netA although -m says FAL_netB: m_model wins (:122)
    torch.save({"epoch": 3, "m_model": "FAL_netA", "state_dict": sd, "best_rmse": 1.0}, ckdir / "checkpoint.pth.tar")
    argv = ["-d", str(root), "-tn", "Kitti2015", "-relbase", "1", "-mdisp", "300", "-mindisp", "2", "-b", "4", "-eval", "True", "-save", "False",
            "-save_pc", "False", "-save_pan", "False", "-save_input", "False", "-w", "1", "--sparse", "-p", "1", "-gpu_no", "0",
            "-dt", str(tmp_path / "Kitti_stage2"), "-ts", "10-18-15_42", "-m", "FAL_netB", "-no_levels", "33",
            "-dtl", ",e20es,b4,lr5e-05/checkpoint.pth.tar", "-fpp", "False", "-mspp", "True", "-median", "False"]
    r = subprocess.run([sys.executable, os.path.join(ROOT, "Test_KITTI.py")] + argv, capture_output=True, text=True, timeout=900, cwd=tmp_path)
    assert r.returncode == 0, r.stderr[-3000:]
    assert "=> using pre-trained model for pan 'FAL_netA'" in r.stdout
    out = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert out["frames"] == 1 and out["post"] == "ms_pp" and out["epe"] > 0
    res = tmp_path / "Test_Results" / "Kitti2015" / "FAL_netB" / "10-18-15_42mspp"
    assert os.path.isfile(res / "errors.txt") and os.path.isfile(res / "settings.txt")
    # the dump switches parse, and are refused (out of scope) when true
    bad = subprocess.run([sys.executable, os.path.join(ROOT, "Test_KITTI.py")] + argv + ["-save", "True"], capture_output=True, text=True, timeout=300, cwd=tmp_path)
    assert bad.returncode != 0 and "out of scope" in (bad.stderr + bad.stdout)


@pytest.mark.parametrize("mode", ["Kitti2015", "Kitti_eigen_test_improved"])
def test_test_kitti_evaluates_a_dataset(mode, tmp_path):
    """Test_KITTI.py as an evaluator (reference Test_KITTI.py:103-117,255-280): file-list dataset at B = 1 over a generated PNG
    tree -> forward + ms_pp in f16 -> KITTI errors (+ EPE for KITTI 2015) -> errors.txt; the same frames through
    inference.evaluate in f32 must give the metrics the CPU oracle's disparities give (the metric chain itself is pinned by G6)."""
    import json
    from PIL import Image
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from test_host_logic import _write_png_fixture
    from fal_net_amd import datasets as DS
    from oracle import falnet_oracle as O
    root, _ = _write_png_fixture(tmp_path, n_train=1, n_val=2)
    args = ["-tn", mode]
    if mode == "Kitti2015":
        troot = os.path.join(root, "Kitti2015")
        triples = DS.kitti2015_pairs(troot)
    else:  # Eigen split: <drive>/image_02/data/<frame>.png + <drive>/proj_depth/groundtruth/image_02/<frame>.png (uint16 depth * 256)
        rng = np.random.default_rng(3)
        troot = os.path.join(root, mode)
        lines = []
        for i in range(2):
            drive = os.path.join("2011_09_26", "2011_09_26_drive_0002_sync")
            for cam in ("image_02", "image_03"):
                os.makedirs(os.path.join(troot, drive, cam, "data"), exist_ok=True)
                Image.fromarray(rng.integers(0, 256, (375, 1242, 3), dtype=np.uint8)).save(os.path.join(troot, drive, cam, "data", f"{i:010d}.png"))
            os.makedirs(os.path.join(troot, drive, "proj_depth", "groundtruth", "image_02"), exist_ok=True)
            depth = (rng.random((375, 1242)) * 80 * 256).astype(np.uint16)
            depth[rng.random((375, 1242)) < 0.7] = 0
            Image.fromarray(depth).save(os.path.join(troot, drive, "proj_depth", "groundtruth", "image_02", f"{i:010d}.png"))
            lines.append(f"{drive}/image_02/data/{i:010d}.png {drive}/image_03/data/{i:010d}.png")
        lines.append("2011_09_26/2011_09_26_drive_0002_sync/image_02/data/0000009999.png 2011_09_26/2011_09_26_drive_0002_sync/image_03/data/0000009999.png")
        lst = tmp_path / "eigen_test.txt"
        lst.write_text("\n".join(lines) + "\n")
        args += ["--test_list", str(lst)]
        triples = DS.eigen_test_triples(str(lst), troot)
    assert len(triples) == 2
    save = tmp_path / "res"
    cmd = [sys.executable, os.path.join(ROOT, "Test_KITTI.py"), "-d", str(root), "--allow-seeded-weights", "--dtype", "f16", "-w", "2",
           "--save-path", str(save)] + args
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-3000:]
    out = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert out["frames"] == 2 and set(out["kitti"]) == set(utils.kitti_error_names) and all(np.isfinite(list(out["kitti"].values())))
    txt = open(save / "errors.txt").read()
    assert "Number of parameters" in txt and "EPE" in txt and "abs_rel" in txt
    if mode == "Kitti2015":
        assert out["epe"] > 0
    # without a checkpoint and without the explicit flag the evaluator refuses (the reference torch.loads the checkpoint)
    r2 = subprocess.run([sys.executable, os.path.join(ROOT, "Test_KITTI.py"), "-d", str(root)] + args, capture_output=True, text=True, timeout=300, cwd=ROOT)
    assert r2.returncode != 0 and "--checkpoint" in (r2.stderr + r2.stdout)
    # f32 evaluation loop vs the oracle's disparities through the same metric chain
    sd = synthetic.seeded_falnetb_state_dict(49)
    m = FAL_netB({"state_dict": sd}, 49, compute_dtype=torch.float32).to(DEV).eval()
    loader = DS.make_loader(DS.StereoValDataset(troot, triples[:1]), 1, 0, shuffle=False, drop_last=False)
    got = inference.evaluate(m, loader, data_name=mode, post="ms_pp", log=None)
    left_u8, _, gt = DS.StereoValDataset(troot, triples[:1])[0]
    left = DS.to_model_input(left_u8, "cpu")
    mx = torch.full((1, 1, 1), 300.0)
    mn = mx * 2.0 / 300.0
    with torch.no_grad():
        ref = O.ms_pp(sd, left, O.falnet_forward(sd, left, mn, mx), mn, mx)
    t_np, p_np = gt.view(1, *gt.shape).numpy(), ref.squeeze(1).numpy()
    gd, pd = (utils.disps_to_depths_kitti2015 if mode == "Kitti2015" else utils.disps_to_depths_kitti)(t_np, p_np)
    want = utils.compute_kitti_errors(gd[0], pd[0])
    for name, w in zip(utils.kitti_error_names, want):
        assert abs(got["kitti"][name] - w) <= 1e-3 * abs(w) + 1e-6, (name, got["kitti"][name], w)
