"""CPU: host-side logic of the hot path (no kernels run): tap tables, channel-group packing math, module surface,
checkpoint keys, loud failure without a GPU."""
import itertools

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from fal_net_amd import ops, synthetic
from fal_net_amd.models import FAL_netB


def test_state_dict_surface_matches_reference_keys():
    m = FAL_netB(None, no_levels=49)
    keys = list(m.state_dict().keys())
    assert keys == list(synthetic.falnetb_param_shapes(49).keys())
    for k, shape in synthetic.falnetb_param_shapes(49).items():
        assert tuple(m.state_dict()[k].shape) == tuple(shape), k
    assert sum(p.numel() for p in m.parameters()) == 16974354  # FAL_netB N=49
    trainable = sum(p.numel() for n, p in m.named_parameters() if "amask_conv" not in n)
    assert trainable == 16932402
    assert len(m.weight_parameters()) + len(m.bias_parameters()) == len(list(m.parameters()))
    m2 = FAL_netB({"state_dict": synthetic.seeded_falnetb_state_dict(33)}, no_levels=33)
    assert m2.conv0.weight.shape == (33, 33, 1, 1)


def test_forward_on_cpu_fails_loudly():
    m = FAL_netB(None, no_levels=7)
    x = torch.zeros(1, 3, 64, 128)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        m(x, torch.ones(1, 1, 1), torch.ones(1, 1, 1) * 30)
    from fal_net_amd import loss_functions as LF
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        LF.vgg(x)


def test_stride2_dgrad_parity_tables_are_exact():
    """The 4 parity-class tap tables reproduce the full transposed conv (checked with torch on CPU)."""
    g = torch.Generator().manual_seed(0)
    for H, W in ((8, 10), (7, 9)):
        x = torch.randn(1, 2, H, W, generator=g, requires_grad=True)
        w = torch.randn(3, 2, 3, 3, generator=g)
        y = F.conv2d(x, w, stride=2, padding=1)
        gy = torch.randn(y.shape, generator=g)
        y.backward(gy)
        OH, OW = y.shape[2:]
        gin = torch.zeros(1, 2, H, W)
        for py, px in itertools.product(range(2), range(2)):
            for dy, dx, wi in ops.dgrad_taps_s2(py, px):
                kh, kw = wi // 3, wi % 3
                for ty in range((H - py + 1) // 2):
                    for tx in range((W - px + 1) // 2):
                        oy, ox = ty + dy, tx + dx
                        if 0 <= oy < OH and 0 <= ox < OW:
                            gin[0, :, 2 * ty + py, 2 * tx + px] += (gy[0, :, oy, ox, None] * w[:, :, kh, kw]).sum(0)
        assert torch.allclose(gin, x.grad, atol=1e-5)
    # every 3x3 tap appears in exactly one parity class
    seen = sorted(wi for py in range(2) for px in range(2) for _, _, wi in ops.dgrad_taps_s2(py, px))
    assert seen == list(range(9))


def test_tap_tables_forward_and_flipped():
    f, d = ops.fwd_taps(3), ops.dgrad_taps_s1(3)
    assert [(dy, dx) for dy, dx, _ in f] == [(kh - 1, kw - 1) for kh in range(3) for kw in range(3)]
    for (dy, dx, wi) in d:  # spatial tap (dy,dx) of a stride-1 dgrad uses packed weight tap 8 - spatial index
        assert wi == 8 - ((dy + 1) * 3 + (dx + 1))
    assert ops.fwd_taps(1) == [(0, 0, 0)]


def test_packed_conv_channel_groups():
    w = torch.nn.Parameter(torch.zeros(64, 33, 3, 3))
    pc = ops.PackedConv("conv1", w, None, [32, 1], stride=2)
    assert pc.groups_pad == [32, 32] and pc.cin_pad == 64 and pc.cout_pad == 64 and pc.group_channels() == (32, 32)
    w = torch.nn.Parameter(torch.zeros(49, 96, 3, 3))
    pc = ops.PackedConv("iconv1", w, None, [64, 32])
    assert pc.cin_pad == 96 and pc.cout_pad == 64
    w = torch.nn.Parameter(torch.zeros(32, 3, 3, 3))
    pc = ops.PackedConv("conv0", w, None, [3])
    assert pc.cin_pad == 32 and pc.group_channels() == (3, 32)
    assert ops.pad_c(49) == 64 and ops.pad_c(96) == 96 and ops.pad_c(1) == 32


def test_synthetic_is_deterministic():
    a, b = synthetic.seeded_falnetb_state_dict(7), synthetic.seeded_falnetb_state_dict(7)
    assert all(torch.equal(a[k], b[k]) for k in a)
    l1, r1, mn1, mx1 = synthetic.synthetic_pair(2, 8, 16, seed=3, distinct=True)
    l2, r2, mn2, mx2 = synthetic.synthetic_pair(2, 8, 16, seed=3, distinct=True)
    assert torch.equal(l1, l2) and torch.equal(r1, r2) and torch.equal(mx1, mx2)
    assert float(mx1[1]) < float(mx1[0]) and torch.allclose(mn1, mx1 * 2 / 300)
    v = synthetic.seeded_vgg19_state_dict()
    assert v["features.16.weight"].shape == (256, 256, 3, 3)


def test_kitti_metrics_against_golden(golden_dir):
    import os
    from fal_net_amd import myUtils as U
    g = np.load(os.path.join(golden_dir, "g6_losses_metrics.npz"))
    assert np.allclose(U.compute_kitti_errors(g["gt"], g["pr"]), g["kitti_errors"], rtol=1e-6)
    td, pd = U.disps_to_depths_kitti2015(g["gd"], g["pd"])
    assert np.allclose(np.asarray(td), g["gt_depth"], rtol=1e-6) and np.allclose(np.asarray(pd), g["pred_depth"], rtol=1e-6)


def _write_png_fixture(tmp_path, n_train=4, n_val=2, val_size=(375, 1242)):
    """A KITTI-shaped tree of generated PNGs (no dataset needed): <root>/Kitti/<drive>/image_0{2,3}/data/*.png + a pair list in the
    format of the reference's Datasets/kitti_eigen_train.txt, and <root>/Kitti2015/training/{image_2,image_3,disp_occ_0}."""
    import numpy as np
    from PIL import Image
    rng = np.random.default_rng(7)
    root = tmp_path / "data"
    lines = []
    for i in range(n_train):
        for cam in ("image_02", "image_03"):
            d = root / "Kitti" / "2011_09_26" / "drive_0001_sync" / cam / "data"
            d.mkdir(parents=True, exist_ok=True)
            Image.fromarray(rng.integers(0, 256, (110 + i, 330 - 2 * i, 3), dtype=np.uint8)).save(d / f"{i:010d}.png")
        lines.append(f"2011_09_26/drive_0001_sync/image_02/data/{i:010d}.png 2011_09_26/drive_0001_sync/image_03/data/{i:010d}.png")
    lines.append("2011_09_26/drive_0001_sync/image_02/data/9999999999.png 2011_09_26/drive_0001_sync/image_03/data/9999999999.png")  # absent: skipped
    lst = tmp_path / "train_pairs.txt"
    lst.write_text("\n".join(lines) + "\n")
    for i in range(n_val):
        for sub in ("image_2", "image_3"):
            d = root / "Kitti2015" / "training" / sub
            d.mkdir(parents=True, exist_ok=True)
            Image.fromarray(rng.integers(0, 256, (*val_size, 3), dtype=np.uint8)).save(d / f"{i:06d}_10.png")
        d = root / "Kitti2015" / "training" / "disp_occ_0"
        d.mkdir(parents=True, exist_ok=True)
        disp = (rng.random(val_size) * 80 * 256).astype(np.uint16)
        disp[rng.random(val_size) < 0.5] = 0  # sparse ground truth
        Image.fromarray(disp).save(d / f"{i:06d}_10.png")
    return root, lst


def test_real_data_loader_on_generated_pngs(tmp_path):
    """File list -> decode in loader workers -> lists of uint8 pairs (frames of different sizes), KITTI-2015 validation triples with
    the uint16 / 256 disparity decoding (reference: Datasets/Kitti.py:37-41, listdataset_train.py:70-98, listdataset_test.py:43-46)."""
    from fal_net_amd import datasets as DS
    root, lst = _write_png_fixture(tmp_path)
    kroot = str(root / "Kitti")
    pairs = DS.read_pair_list(str(lst), kroot)
    assert len(pairs) == 4  # the absent pair is dropped like Kitti.py:40-41 does
    ds = DS.StereoPairDataset(kroot, pairs, max_pix=300, fix=True)
    seen = 0
    for batch in DS.make_loader(ds, batch_size=2, workers=2, shuffle=True):
        assert isinstance(batch, list) and len(batch) == 2
        for left, right, x_pix in batch:
            assert left.dtype == torch.uint8 and left.dim() == 3 and left.shape[2] == 3 and left.shape == right.shape and x_pix == 300.0
            seen += 1
    assert seen == 4
    swapped = {DS.StereoPairDataset(kroot, pairs, max_pix=300, fix=False)[0][2] for _ in range(40)}
    assert swapped == {300.0, -300.0}  # random view order with the sign of the disparity range (listdataset_train.py:70-79)
    tr = DS.kitti2015_pairs(str(root / "Kitti2015"))
    assert len(tr) == 2
    left, right, disp = DS.StereoValDataset(str(root / "Kitti2015"), tr)[1]
    assert left.shape == (375, 1242, 3) and disp.shape == (375, 1242) and disp.dtype == torch.float32
    assert 0.0 <= float(disp.min()) and float(disp.max()) < 80.0 and float((disp == 0).float().mean()) > 0.3
    x = DS.to_model_input(left, "cpu")
    assert x.shape == (1, 3, 375, 1242) and abs(float(x[0, 1].mean()) - (float(left[..., 1].float().mean()) / 255 - 0.432)) < 1e-5
    with pytest.raises(FileNotFoundError):
        DS.read_pair_list(str(tmp_path / "nope.txt"), kroot)


def test_test_kitti_accepts_the_reference_command_line():
    """Every flag of the reference's parser (reference Test_KITTI.py:36-60) parses with its meaning: -m is the model name, the checkpoint
    is <-dt>/<-ts>/<-m><-dtl> (:119-120), the booleans take a value, the dump switches are refused when true."""
    import importlib.util
    import os
    spec = importlib.util.spec_from_file_location("Test_KITTI_entry", os.path.join(os.path.dirname(__file__), "..", "Test_KITTI.py"))
    tk = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(tk)
    d = tk.parser.parse_args([])  # the reference's defaults
    assert (d.tdataName, d.max_disp, d.min_disp, d.batch_size, d.evaluate, d.save, d.save_pc, d.save_pan, d.save_input, d.workers, d.sparse,
            d.print_freq, d.dataset, d.time_stamp, d.model, d.no_levels, d.details, d.f_post_process, d.ms_post_process, d.median, d.rel_baselne) == \
        ("Kitti_eigen_test_improved", 300, 2, 1, True, False, False, False, False, 4, False, 10, "Kitti_stage2", "10-18-15_42", "FAL_netB", 49,
         ",e20es,b4,lr5e-05/checkpoint.pth.tar", False, True, False, 1)
    assert tk.checkpoint_path(d) == os.path.join("Kitti_stage2", "10-18-15_42", "FAL_netB,e20es,b4,lr5e-05/checkpoint.pth.tar")
    a = tk.parser.parse_args(["-d", "/data", "-tn", "Kitti2015", "-relbase", "0.5", "-mdisp", "192", "-mindisp", "1", "-b", "2", "-eval", "False",
                              "-save", "False", "-save_pc", "False", "-save_pan", "False", "-save_input", "False", "-w", "2", "--sparse", "-p", "5",
                              "-gpu_no", "3", "-dt", "Kitti_stage1", "-ts", "01-02-03_04", "-m", "FAL_netC", "-no_levels", "33", "-dtl", "/model_best.pth.tar",
                              "-fpp", "True", "-mspp", "False", "-median", "True"])
    assert (a.max_disp, a.min_disp, a.rel_baselne, a.evaluate, a.f_post_process, a.ms_post_process, a.median, a.gpu_no, a.model) == \
        (192.0, 1.0, 0.5, False, True, False, True, "3", "FAL_netC")
    assert tk.checkpoint_path(a) == os.path.join("Kitti_stage1", "01-02-03_04", "FAL_netC/model_best.pth.tar")
    tk.refuse_out_of_scope(a)
    assert tk.checkpoint_path(tk.parser.parse_args(["--checkpoint", "x.pth.tar"])) == "x.pth.tar"
    for flag in ("-save", "-save_pc", "-save_pan", "-save_input"):
        with pytest.raises(SystemExit, match="out of scope"):
            tk.refuse_out_of_scope(tk.parser.parse_args([flag, "True"]))
    with pytest.raises(SystemExit):  # the round-3 spellings are gone: -m is no longer a path, -maxd never was a reference flag
        tk.parser.parse_args(["-maxd", "300"])
