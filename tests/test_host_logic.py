"""CPU tests of the host side: the C-ABI library loads and exports everything
include/fnssl.h declares; host-only entry points (packing, coefficient table,
shape helpers, argument validation) behave; and the packed LSTM weight stream,
executed by a lane-level emulation of the kernel's MFMA data flow, reproduces the
oracle.  No compute call touches a GPU here.
"""
import ctypes as C
import os
import re

import numpy as np
import pytest

from conftest import ROOT, assert_close, rs_randn
from fnssl import _lib, ops
from fnssl import weights as W
from oracle import fnssl_oracle as O
from wave_emulator import run_wave


def test_library_exports_every_declared_symbol():
    lib = _lib.load()
    header = open(os.path.join(ROOT, "include", "fnssl.h")).read()
    declared = sorted(set(re.findall(r"\b(fnssl_[a-z0-9_]+)\s*\(", header)))
    assert declared, "no declarations parsed"
    for name in declared:
        assert hasattr(lib, name), "library does not export %s" % name
    assert sorted(_lib.SYMBOLS) == declared
    assert lib.fnssl_abi_version() == _lib.ABI_VERSION == 19


def test_shape_helpers_match_reference_formulas():
    for ns in (512, 767, 768, 64000, 77056, 6400):
        assert ops.num_frames(ns) == O.n_frames(ns)
    assert ops.num_frames(100) == 0
    assert ops.num_pairs(4, "MM") == 6 and ops.num_pairs(4, "M") == 3 and ops.num_pairs(2, "MM") == 1
    assert ops.num_pairs(1, "MM") == 0


def test_forgetting_coefs_bitwise():
    for nt, sl in ((24, 8), (300, 298), (310, 298), (5, 298)):
        a, b = ops.forgetting_coefs_host(nt, sl)
        ra, rb = O.forgetting_coefs(nt, sl)
        np.testing.assert_array_equal(a, ra)
        np.testing.assert_array_equal(b, rb)


def test_pack_rejects_bad_sizes():
    lib = _lib.load()
    assert lib.fnssl_lstm_packed_floats(4, 0, 128) == 8 * (1 + 1 + 8) * 4 * 256
    assert lib.fnssl_lstm_packed_floats(256, 4, 256) == 16 * (1 + 16 + 1 + 16) * 4 * 256
    assert lib.fnssl_lstm_packed_floats(3, 0, 128) == 0      # not a multiple of 4
    assert lib.fnssl_lstm_packed_floats(4, 0, 24) == 0       # hidden not a multiple of 16
    with pytest.raises(RuntimeError):
        ops.pack_lstm_host(np.zeros((64, 6), np.float32), np.zeros((64, 16), np.float32),
                           np.zeros(64, np.float32), np.zeros(64, np.float32), 6, 0)


def test_lstm_forward_validates_before_touching_the_device():
    lib = _lib.load()
    d = _lib.LstmDesc()
    d.hidden = 100
    assert lib.fnssl_lstm_forward(C.byref(d), None) == -1
    assert b"hidden size" in lib.fnssl_last_error()
    with pytest.raises(RuntimeError, match="hidden size"):
        _lib.check(-1, "lstm_forward")


def test_ops_refuse_cpu_tensors():
    import torch
    with pytest.raises(RuntimeError, match="ROCm device tensor"):
        ops.stft(torch.zeros(1, 1024, 2))
    with pytest.raises(RuntimeError, match="ROCm device tensor"):
        ops.head(torch.zeros(1, 4, 12, 256), torch.zeros(2, 256), torch.zeros(2))


@pytest.mark.parametrize("c0,c2,H,reverse", [(4, 0, 16, False), (32, 4, 32, False), (16, 0, 16, True),
                                             (36, 0, 16, False), (8, 20, 16, True)])
def test_packed_stream_through_mfma_emulation_matches_oracle(c0, c2, H, reverse):
    I = c0 + c2
    sd = W.make_state([("L." + n, s) for n, s in W.lstm_param_shapes(I, H, False)], seed=4000 + I + H)
    T = 5
    x = rs_randn(4100 + I, (16, T, I))
    packed = ops.pack_lstm_host(sd["L.weight_ih_l0"], sd["L.weight_hh_l0"], sd["L.bias_ih_l0"],
                                sd["L.bias_hh_l0"], c0, c2)
    got = run_wave(packed, x[:, :, :c0] if c0 else None, x[:, :, c0:] if c2 else None, H, T, reverse)
    want = O.lstm_dir(x, sd["L.weight_ih_l0"], sd["L.weight_hh_l0"], sd["L.bias_ih_l0"], sd["L.bias_hh_l0"],
                      reverse=reverse)
    assert_close(got, want, 1e-5, 1e-6, "emulated wave vs oracle")


def test_checkpoint_key_names_match_reference():
    """state_dict keys/shapes are the drop-in contract; CPU-only check of the module."""
    import Model as at_model
    from fnssl import weights as W
    for online in (True, False):
        sd = at_model.FN_SSL(is_online=online).state_dict()
        want = dict(W.fnssl_param_shapes(is_online=online))
        assert {k: tuple(v.shape) for k, v in sd.items()} == want
    assert list(at_model.FN_lightning().state_dict())[0].startswith("arch.block_1.fullLstm.")


def test_ipdnet_dropin_state_dict_matches_reference_names():
    """Checkpoint compatibility of the IPDnet drop-in: parameter names and shapes are the reference's
    (weights.ipdnet_param_shapes was checked against the real reference when g10 was generated)."""
    import importlib.util
    import os
    from conftest import ROOT
    from fnssl import weights as W
    spec = importlib.util.spec_from_file_location(
        "fnssl_ipdnet_dropin_cpu", os.path.join(ROOT, "fn-ssl_amd", "IPDnet", "FixedAarryIPDnet.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    for isz, hid, mt, online in [(4, 128, 2, True), (16, 256, 2, True), (8, 256, 3, False)]:
        net = mod.IPDnet(input_size=isz, hidden_size=hid, max_track=mt, is_online=online)
        got = {k: tuple(v.shape) for k, v in net.state_dict().items()}
        want = {k: tuple(s) for k, s in W.ipdnet_param_shapes(isz, hid, mt, online)}
        assert got == want
    with __import__("pytest").raises(RuntimeError, match="eval"):
        net(__import__("torch").zeros(1, 8, 4, 12))


def test_spatialnet_entry_points_validate_before_touching_the_device():
    """fnssl_sn_* (IPDnet2 row): bad shapes / null pointers are refused with a message, no launch."""
    lib = _lib.load()
    v = _lib.BtfView(0, 0, 0, 0)
    w = _lib.SnFconvW()
    assert lib.fnssl_sn_fconv(C.byref(v), 1, 1, 16, C.byref(w), 1, 1, None, 0, 0, 0, 0, None) == -1
    assert b"aligned" in lib.fnssl_last_error() or b"null" in lib.fnssl_last_error()
    buf = np.zeros(64, np.float32)
    v = _lib.BtfView(buf.ctypes.data // 16 * 16, 96, 96, 96)
    for name in ("ln_w", "ln_b", "wT", "bias", "prelu"):
        setattr(w, name, buf.ctypes.data)
    assert lib.fnssl_sn_fconv(C.byref(v), 1, 1, 24, C.byref(w), 1, 1, v.p, 96, 96, 96, 0, None) == -1   # nf not 2^k
    assert b"power of two" in lib.fnssl_last_error()
    assert lib.fnssl_sn_fconv(C.byref(v), 1, 1, 16, C.byref(w), 1, 3, v.p, 96, 96, 96, 0, None) == -1   # pool 3
    assert lib.fnssl_sn_fconv(C.byref(v), 1, 1, 16, C.byref(w), 1, 1, v.p, 96, 96, 96, 5, None) == -1   # precision 5
    assert b"precision" in lib.fnssl_last_error()
    assert lib.fnssl_sn_mamba_workspace_bytes(2, 10, 16) == 2 * 10 * 16 * (384 + 40 + 192) * 4
    assert lib.fnssl_sn_forward_workspace_bytes(0, 256, 10) == 0
    net = _lib.SnNet()
    net.dim_input, net.num_layers, net.time_ratio = 10, 8, 5
    assert lib.fnssl_sn_state_floats(C.byref(net), 2, 256) == 2 * 10 * 256 * 4 + 8 * 2 * 32 * (3 * 192 + 192 * 16)
    assert lib.fnssl_sn_forward(C.byref(net), buf.ctypes.data, 0, 0, 0, 1, 1, 200, 10, None, 0, buf.ctypes.data, None, 0,
                                None) == -1
    assert b"num_freqs" in lib.fnssl_last_error()
    net.precision = 3
    assert lib.fnssl_sn_forward(C.byref(net), buf.ctypes.data, 0, 0, 0, 1, 1, 256, 10, None, 0, buf.ctypes.data, None, 0,
                                None) == -1
    assert b"precision" in lib.fnssl_last_error()


def test_ipdnet2_dropin_keeps_reference_names_and_rejects_cpu():
    import importlib.util
    import torch
    spec = importlib.util.spec_from_file_location("fnssl_ipdnet2_dropin_cpu",
                                                  os.path.join(ROOT, "fn-ssl_amd", "IPDnet2", "IPDnet2.py"))
    M = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(M)
    net = M.OnlineSpatialNet(dim_input=10, dim_output=16, num_layers=2, dim_hidden=96, num_heads=4, dim_squeeze=8,
                             num_freqs=256, attention="mamba(16,4)", time_compression_layer=0).eval()
    names = {k: tuple(v.shape) for k, v in net.state_dict().items()}
    assert names == dict(W.ipdnet2_param_shapes(num_layers=2))   # = the reference's keys (G14 was loaded strictly)
    with pytest.raises(RuntimeError):
        net(torch.zeros(1, 10, 256, 10))                         # no CPU path
    with pytest.raises(NotImplementedError):
        M.OnlineSpatialNet(dim_input=10, dim_output=16, num_layers=2, dim_hidden=96, dim_squeeze=8, num_freqs=256,
                           attention="mhsa(251)")
    # .bfloat16() is what selects FNSSL_PRECISION_BF16 (BASELINE config 5), for the network and for every layer
    from fnssl import spatialnet as sn
    assert M._prec(net) == sn.FP32 == 0 and M._prec(net.layers[0]) == sn.FP32
    net = net.bfloat16()
    assert M._prec(net) == sn.BF16 == 1 and M._prec(net.layers[1]) == sn.BF16 and M._prec(net.encoder) == sn.BF16
    assert _lib.SnNet.precision.offset > _lib.SnNet.bd.offset         # the field the C side reads last in fnssl_sn_net


def test_predict_cli_accepts_the_reference_flag_surface():
    """Opt.py:21-43: every flag of the reference parses with its type / default and maps onto the path."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("fnssl_predict_cli", os.path.join(ROOT, "fn-ssl_amd", "Predict.py"))
    P = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(P)
    a = P.parse_args(["--test", "--gpu-id", "1,2", "--workers", "4", "--use-amp", "--seed", "7", "--checkpoint-start",
                      "--time", "04231627", "--sources", "1", "2", "--source-state", "static", "--localize-mode", "IDL",
                      "unkNum", "2", "--bz", "2", "2", "8", "--epochs", "3", "--lr", "0.01", "--datasetMode", "locata"])
    assert (a.gpu_id, a.workers, a.use_amp, a.seed, a.bz, a.localize_mode) == ("1,2", 4, True, 7, [2, 2, 8],
                                                                              ["IDL", "unkNum", 2])
    d = P.parse_args([])                                          # defaults of Opt.py (+ the implied --test)
    assert (d.gpu_id, d.bz, d.sources, d.localize_mode, d.lr, d.epochs, d.datasetMode, d.test) == \
        ("0,1", [1, 1, 1], [1], ["IDL", "kNum", 1], 0.001, 100, "simulate", True)
    with pytest.raises(Exception, match="Stage of train or test"):   # Opt.py:48-49
        P.parse_args(["--train", "--test"])
    with pytest.raises(SystemExit, match="no CPU implementation"):
        P.main(["--test", "--no-cuda", "--synthetic", "1"])
    with pytest.raises(SystemExit, match="training_step"):
        P.main(["--train"])


def test_lightning_module_uses_manual_optimization(monkeypatch):
    """With pytorch_lightning present MyModel is a LightningModule whose training_step runs the whole optimisation
    step in the HIP engine and returns a detached loss: automatic optimisation must be off (Lightning 2.x would
    otherwise call backward() on that loss), and — because Lightning's manual-optimisation loop counts OPTIMIZER steps —
    every training_step must step the optimizer Lightning wraps, or trainer.global_step stays 0 and ModelCheckpoint
    (which skips saving while global_step == its last saved step, initially 0), max_steps and logger indices never
    move.  Lightning is absent from the image, so stubs with that bookkeeping stand in for it."""
    import importlib
    import sys
    import types
    import torch
    pl = types.ModuleType("pytorch_lightning")

    class LightningOptimizer:                      # Lightning's wrapper: counts completed optimizer steps
        def __init__(self, opt, trainer):
            self._optimizer, self._trainer = opt, trainer

        def step(self, closure=None):
            out = self._optimizer.step(closure)
            self._trainer.optim_step_completed += 1
            return out

    class Trainer:                                 # global_step as Lightning 2.x derives it under manual optimisation
        def __init__(self, module):
            self.optim_step_completed, self.saved, self._last_global_step_saved = 0, [], 0
            self.module = module
            module._trainer = self
            self._opts = [LightningOptimizer(module.configure_optimizers(), self)]

        @property
        def global_step(self):
            return self.optim_step_completed

        def maybe_checkpoint(self):                # ModelCheckpoint._should_skip_saving_checkpoint
            if self._last_global_step_saved == self.global_step:
                return
            self.saved.append({"global_step": self.global_step,
                               "optimizer_states": [o._optimizer.state_dict() for o in self._opts]})
            self._last_global_step_saved = self.global_step

    class LightningModule(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.automatic_optimization = True
            self._trainer = None

        def optimizers(self):
            return self._trainer._opts[0]

    pl.LightningModule = LightningModule
    monkeypatch.setitem(sys.modules, "pytorch_lightning", pl)
    monkeypatch.delitem(sys.modules, "predict_step", raising=False)
    ps = importlib.import_module("predict_step")
    try:
        m = ps.MyModel(device="cpu")
        assert isinstance(m, LightningModule) and m.automatic_optimization is False
        opt = m.configure_optimizers()
        assert isinstance(opt, torch.optim.Optimizer) and isinstance(opt, ps.EngineOptimizer)

        class FakeEngine:                          # stands in for fnssl.train.TrainEngine (which needs the GPU)
            def __init__(self):
                self.exp_avg, self.exp_avg_sq = torch.zeros(5), torch.zeros(5)
                self.step_count, self.lr = 0, 1e-3

            def equal_shard_pair_offset(self, nbp):    # rank * pairs: no collective, no host sync per step
                return 0

            def step(self, x, gt, sync_loss=True, pair_offset=None):
                assert pair_offset == 0, "the drop-in training_step passes the offset (no all_gather + .item() per step)"
                self.step_count += 1
                self.exp_avg += 1.0
                return torch.tensor([0.25 / self.step_count])

        m._train_engine = FakeEngine()
        monkeypatch.setattr(ps.ops, "preprocess", lambda sig, *a, **k: sig)
        tr = Trainer(m)
        batch = (torch.zeros(1, 4, 2), {"ipd": torch.zeros(1, 1, 512, 1)})
        for i in range(3):
            out = m.training_step(batch, i)
            assert out["loss"].shape == () and not out["loss"].requires_grad
            assert tr.global_step == i + 1                       # the counter Lightning checkpoints / stops on
            tr.maybe_checkpoint()
        assert [c["global_step"] for c in tr.saved] == [1, 2, 3]     # a checkpoint IS written once steps advance
        # the engine's Adam state rides in the optimizer state and restores into a fresh module
        st = tr.saved[-1]["optimizer_states"][0]
        assert st["steps_taken"] == 3 and st["engine"]["step_count"] == 3 and float(st["engine"]["exp_avg"][0]) == 3.0
        m2 = ps.MyModel(device="cpu")
        m2._train_engine = FakeEngine()
        m2.configure_optimizers().load_state_dict(st)
        assert m2._train_engine.step_count == 3 and float(m2._train_engine.exp_avg[0]) == 3.0
        # without a trainer (plain nn.Module use, Predict.py --train) training_step must not need one
        m3 = ps.MyModel(device="cpu")
        m3._train_engine = FakeEngine()
        assert m3.training_step(batch, 0)["loss"].shape == ()
        # fused_engine=False = the reference's route (main.py:149-157, 269-279): AUTOMATIC optimisation (Lightning calls
        # backward() on the returned loss, which carries the graph of fnssl.autograd), torch Adam + ExponentialLR
        m4 = ps.MyModel(device="cpu", fused_engine=False)
        assert m4.automatic_optimization is True
        cfg = m4.configure_optimizers()
        assert isinstance(cfg["optimizer"], torch.optim.Adam) and cfg["optimizer"].defaults["lr"] == 0.001
        assert isinstance(cfg["lr_scheduler"]["scheduler"], torch.optim.lr_scheduler.ExponentialLR)
        assert cfg["lr_scheduler"]["scheduler"].gamma == 0.8988 and cfg["lr_scheduler"]["monitor"] == "valid/loss"
        assert {id(p) for g in cfg["optimizer"].param_groups for p in g["params"]} == {id(p) for p in m4.arch.parameters()}
    finally:
        sys.modules.pop("predict_step", None)


def test_bench_gpus_n_refuses_to_run_with_fewer_devices():
    """`bench.py --gpus N` outside torch.distributed.run starts N ranks itself; with fewer than N devices it must exit
    non-zero (never a silent 1-rank run that claims nothing about scaling).  No GPU here -> rc 2, no JSON line."""
    import subprocess
    import sys
    env = dict(os.environ)
    env.pop("WORLD_SIZE", None)
    env.pop("RANK", None)
    res = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2"], env=env, stdout=subprocess.PIPE,
                         stderr=subprocess.PIPE, text=True, timeout=300)
    assert res.returncode == 2, (res.returncode, res.stderr[-500:])
    assert res.stdout.strip() == ""
    assert "refusing to run with fewer ranks" in res.stderr
    # a WORLD_SIZE that disagrees with --gpus is an error too (the line's n_gpus must be what was asked for)
    env["WORLD_SIZE"] = "2"
    res = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1"], env=env, stdout=subprocess.PIPE,
                         stderr=subprocess.PIPE, text=True, timeout=300)
    assert res.returncode != 0 and "WORLD_SIZE=2" in res.stderr


def _stub_bench_line(n_nested=7, prose=600):
    """A full bench record shaped like run_workload()'s, with every prose field blown up to `prose` characters and a
    40-kernel table per configuration: far larger than anything a real run produces."""
    long = "x" * prose

    def one(cfg):
        return {"metric": "TF-frames/sec DP-IPD forward, 4-mic 257-bin x 300-frame " + long[:80], "value": 17766.12, "unit": "frames/s",
                "n_gpus": 1, "steps": 20, "warmup": 5, "ms_per_step": 540.312, "higher_is_better": True, "scaling": "weak",
                "vs_baseline": None, "dtype": "f32", "data": "synthetic",
                "config": {"workload": "BASELINE configs[%s]: %s" % (cfg, long), "utterances_per_gpu": 32, "global_batch": 32, "mics": 4,
                           "pairs_per_utterance": 6, "frames": 300, "bins": 257, "parallelism": "dp1 " + long, "chunk_pairs": 0,
                           "gflop_per_frame": 7.676, "stream_chunk_frames": 0},
                "roofline": {"bound": "mfma", "kernel": "lstm_static3_kernel<H=256> " + long, "achieved": 137.01, "peak": 157.3,
                             "unit": "TFLOP/s", "frac": 0.871, "traffic": 316911581653.3333, "traffic_source": long, "launches": 60,
                             "avg_ms": 113.146, "flop_per_launch": 15502147584000.0, "peak_measured": 155.2,
                             "frac_of_peak_measured": 0.8827, "peak_measured_how": long, "peak_measured_sustained": 151.3,
                             "slowest_xcd_mhz": 2238.1, "fastest_xcd_mhz": 2381.9,
                             "peak_sustained_detail": {"xcd_mhz": {str(i): 2300.0 + i for i in range(8)}}},
                "cpu_baseline": {"value": 47.48, "unit": "frames/s", "cores": 16, "kind": "port", "sample": long},
                "parity": {"max_abs_err": 3.725290298461914e-08, "rtol": 1e-4, "atol": 1e-5, "sample": long, "ok": True,
                           "gradients": {"tensors": 38, "worst": long}},
                "ms_per_step_median_hip_events": 540.3, "ms_per_step_per_rank": [540.312], "rccl_world_size": 1, "backend": None,
                "cluster_fallbacks": 0, "peak_mem_gb": 84.8, "whole_path_tflops": 136.4,
                "kernels": {"kernel_%02d" % i: {"ms_per_step": 1.234, "launches_per_step": 3.0, "tflops": 12.34} for i in range(40)},
                "kernels_source": long,
                "frontend": {"bound": "hbm", "kernel": long, "achieved": 1850.6, "peak": 8000.0, "unit": "GB/s", "frac": 0.2313,
                             "traffic": None, "ms_per_step": 0.1487, "algorithmic_bytes_per_step": 275251200},
                "ab": {"leg_%d_%s" % (i, long[:50]): {"A_ms_per_step": [1.0, 2.0], "B_ms_per_step": [1.0, 2.0], "env_B": {"K": "1"},
                                                     "order": long, "gain_of_default_pct": 1.06} for i in range(3)}}

    line = one(1)
    line["other_configs"] = {k: one(k) for k in ["2M", "2off", "2b1", "2s", "3", "4", "5", "6", "7", "8"][:n_nested]}
    if n_nested:
        line["other_configs"]["4"]["error"] = None
        line["other_configs"][list(line["other_configs"])[0]] = {"error": "RuntimeError('" + long + "')"}
    return line


def test_bench_stdout_line_is_compact_and_carries_the_contract(tmp_path):
    """The driver keeps ~10 KB of bench.py's stdout: round 5's 26 KB line was cut mid-way and parsed as null.  compact_line()
    must give < 4 KB that json-round-trips for ANY full record — here one with every prose field blown up and 7 (and 10)
    nested configurations — and keep the contract keys, the roofline / cpu_baseline numbers and five numbers per nested
    configuration; the full record goes to the detail file untouched.  Also on round 5's real 26 KB record."""
    import importlib.util
    import json
    spec = importlib.util.spec_from_file_location("fnssl_bench_mod", os.path.join(ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    fulls = [_stub_bench_line(7), _stub_bench_line(10, prose=3000), _stub_bench_line(0)]
    real = os.path.join(ROOT, "profiles", "r05", "o2_bench_default_steps20_last_build_slowest_box.json")
    fulls.append(json.loads(open(real).read()))
    assert len(json.dumps(fulls[-1])) > 20000
    for full in fulls:
        path = bench.write_detail(full, str(tmp_path / "sub" / "bench_detail.json"))
        assert path and json.load(open(path)) == json.loads(json.dumps(full))
        small = bench.compact_line(full, "gpurun_out/bench_detail.json")
        text = json.dumps(small)
        assert len(text) < 4096 and "\n" not in text, len(text)
        back = json.loads(text)
        for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
                  "dtype", "data"):
            assert back[k] == full[k], k
        assert back["config"]["workload"].startswith("BASELINE configs[")
        r, fr = back["roofline"], full["roofline"]
        for k in ("bound", "achieved", "peak", "unit", "frac", "traffic", "launches", "avg_ms", "flop_per_launch", "peak_measured"):
            assert r[k] == fr[k], k
        assert "traffic_source" not in r and "peak_measured_how" not in r
        c = back["cpu_baseline"]
        assert (c["value"], c["unit"], c["cores"], c["kind"]) == tuple(full["cpu_baseline"][k] for k in ("value", "unit", "cores", "kind"))
        assert isinstance(c["sample"], str) and c["sample"]
        assert back["parity"]["ok"] is True and back["parity"]["max_abs_err"] == full["parity"]["max_abs_err"]
        assert back["cluster_fallbacks"] == 0 and back["rccl_world_size"] == 1
        assert "kernels" not in back and "ab" not in back
        for key, o in (full.get("other_configs") or {}).items():
            b = back["other_configs"][key]
            if "error" in o and o.get("error"):
                assert "error" in b
                continue
            assert b["value"] == o["value"] and b["ms_per_step"] == o["ms_per_step"]
            if len(full["other_configs"]) <= 8:
                assert b["roofline_frac"] == o["roofline"]["frac"] and b["cpu_baseline"] == o["cpu_baseline"]["value"]
                assert b["parity_ok"] is True
    # the peak_measured_sustained pair survives when present
    small = bench.compact_line(fulls[0])
    assert small["roofline"]["peak_measured_sustained"] == 151.3 and small["roofline"]["slowest_xcd_mhz"] == 2238.1


def test_launch_planner_picks_the_cheapest_rounds():
    """The launch planner of fnssl_lstm_forward (host logic, queried without a GPU): a round costs one wave-time per
    wave on its fullest SIMD, so config 2's full-band layers (57 600 sequences x 2 directions = 7 200 wave tasks on 256
    CUs, 29 waves per CU) run 15 + 14, 191 of its 192 pairs (28 waves per CU) run 16 + 12, the narrow-band layers
    (49 152 sequences, 12 waves per CU at 3 per SIMD) one round of 12, and small launches one round."""
    lib = _lib.load()

    def rounds(hidden, nseq, ndir, ncu=256):
        buf = (C.c_int * 16)()
        n = lib.fnssl_lstm_plan_rounds(hidden, nseq, ndir, ncu, buf, 16)
        assert n > 0, lib.fnssl_last_error()
        return [buf[i] for i in range(n)]

    assert rounds(128, 192 * 300, 2) == [15, 14]
    assert rounds(128, 191 * 300, 2) == [16, 12]
    assert rounds(256, 192 * 256, 1) == [12]
    assert rounds(256, 384 * 256, 1) == [12, 12]
    assert rounds(128, 96 * 300, 2) == [15]                     # 3 600 tasks = 14.06 waves per CU
    assert rounds(128, 16 * 300, 2) == [4]                      # (small launches take the split kernels before the planner)
    for nseq in (1, 999, 57600, 123456):                        # every plan covers the work
        for hidden, ndir in ((128, 2), (256, 1)):
            assert sum(rounds(hidden, nseq, ndir)) * 256 >= ((nseq + 15) // 16) * ndir
    assert lib.fnssl_lstm_plan_rounds(64, 100, 1, 256, (C.c_int * 4)(), 4) < 0


def test_lstm_workspace_covers_the_cluster_hand_off_area():
    """fnssl_lstm_workspace_bytes (host logic): hidden 128 / 256 workspaces hold the hand-off area of the cluster-resident
    bf16 kernels — a status word, 1 KiB of tags and two parities of operand records per cluster of 512 (H = 256:
    2 x 2 parts x 8 tiles x 16 KiB) or 768 (H = 128: 2 x 3 x 8 x 8 KiB) sequences per direction; other hidden sizes none."""
    lib = _lib.load()
    ws = lib.fnssl_lstm_workspace_bytes
    per256, per128 = 1024 + 2 * 2 * 8 * 16 * 1024, 1024 + 2 * 3 * 8 * 8 * 1024
    assert ws(16384, 256, 1) - ws(16384 - 512, 256, 1) >= per256            # one cluster more
    assert ws(16384, 256, 1) >= 32 * per256
    assert ws(19200, 128, 2) >= 50 * per128                                  # config 3's full-band layers: 25 clusters x 2
    assert ws(768, 128, 1) - ws(767, 128, 1) < per128                        # 767 and 768 sequences: one cluster each
    assert ws(769, 128, 1) - ws(768, 128, 1) >= per128
    assert ws(16384, 64, 1) < 16384 // 16 * 2 * 4 * 1024                     # H = 64: cell state only
    assert ws(0, 256, 1) == 0


def test_train_mode_forward_has_no_cpu_fallback_and_flat_layout_matches_named_parameters():
    """The autograd route (fnssl/autograd.py) is HIP-only like everything else: a CPU tensor in train mode raises instead
    of silently running nn.LSTM; the flat layout / layer table it shares with TrainEngine follows named_parameters."""
    import torch
    import Model
    from fnssl import autograd as ag
    from fnssl import train
    net = Model.FN_SSL().train()
    with pytest.raises(RuntimeError, match="ROCm device"):
        net(torch.zeros(1, 4, 16, 24))
    blk = Model.FNblock(256, is_online=True).train()
    with pytest.raises(RuntimeError, match="ROCm device"):
        blk(torch.zeros(1, 2, 3, 256), None, torch.zeros(2, 3, 256))
    named = [(k, tuple(p.shape)) for k, p in net.named_parameters()]
    offset, total = train.flat_layout(named)
    assert total == 1 + sum(p.numel() for p in net.parameters()) == 1 + 2511362
    off = 1
    for k, shape in named:
        assert offset[k] == (off, shape)
        off += int(np.prod(shape))
    layers = train.build_layers(True)
    assert [(L.name, L.mode, L.hidden, L.ndir, L.c0, L.c2, L.c0g) for L in layers[:2]] == \
        [("block_1.fullLstm", "full", 128, 2, 4, 0, 0), ("block_1.narrLstm", "narrow", 256, 1, 256, 4, 256)]
    assert [L.ndir for L in train.build_layers(False)] == [2] * 6
    assert ag.base_seed(5, 2) == (5 * 1000003 + 2 * 8191) & 0xFFFFFFFF
    net.force_dropout_base = 77
    assert ag._next_base(net) == 77 and net.dropout_calls == 1
    # the forward-only drop-ins still refuse train mode
    from IPDnet.FixedAarryIPDnet import IPDnet
    with pytest.raises(RuntimeError, match="eval"):
        IPDnet().train()(torch.zeros(1, 4, 256, 24))


def test_library_reads_no_environment_and_tuning_is_explicit(monkeypatch):
    """include/fnssl.h: "the library reads NO environment variable".  (a) the shipping .so does not even import getenv;
    (b) fnssl_tuning round-trips through the ABI, rejects a struct of another size, names every knob the header defines;
    (c) FNSSL_<KNOB> variables are parsed on the PYTHON side (fnssl/_lib.py) — at load and on refresh_tuning() — with the
    old switches' semantics (presence flags, member + 1 for the fault-injection hook); (d) a worker thread sees the
    process default (PyTorch's autograd engine runs backward() on its own thread)."""
    import ctypes as C
    import re
    import subprocess
    import threading
    from fnssl import _lib
    lib = _lib.load()
    und = subprocess.run(["nm", "-D", "--undefined-only", _lib.LIB_PATH], stdout=subprocess.PIPE, text=True, check=True).stdout
    assert "getenv" not in und and "secure_getenv" not in und
    hdr = open(os.path.join(ROOT, "include", "fnssl.h")).read()
    declared = {m.group(1): int(m.group(2)) for m in re.finditer(r"#define FNSSL_TUNE_([A-Z0-9_]+) (\d+)", hdr)}
    count = declared.pop("COUNT")
    assert count == _lib.TUNE_COUNT and C.sizeof(_lib.Tuning) == 4 + 4 * count
    assert _lib.tuning_names() == declared and lib.fnssl_tuning_name(len(declared)) is None
    # (b)
    t = _lib.make_tuning(no_static3=1, cluster_spin_limit=20000)
    assert lib.fnssl_tuning_set(C.byref(t)) == 0
    back = _lib.Tuning()
    assert lib.fnssl_tuning_get(C.byref(back)) == 0
    assert back.knob[declared["NO_STATIC3"]] == 1 and back.knob[declared["CLUSTER_SPIN_LIMIT"]] == 20000
    bad = _lib.Tuning()
    bad.struct_bytes = 12
    assert lib.fnssl_tuning_set(C.byref(bad)) != 0 and b"struct_bytes" in lib.fnssl_last_error()
    with pytest.raises(KeyError):
        _lib.make_tuning(no_such_knob=1)
    # (d)
    seen = {}

    def worker():
        w = _lib.Tuning()
        lib.fnssl_tuning_get(C.byref(w))
        seen["v"] = w.knob[declared["NO_STATIC3"]]

    th = threading.Thread(target=worker)
    th.start()
    th.join()
    assert seen["v"] == 1
    # (c)
    for k in list(os.environ):
        if k.startswith("FNSSL_"):
            monkeypatch.delenv(k)
    _lib.refresh_tuning()
    lib.fnssl_tuning_get(C.byref(back))
    assert not any(back.knob)
    env = {"FNSSL_NO_F32_CLUSTER": "0", "FNSSL_CLUSTER_TEST_STALL": "3", "FNSSL_LSTM_VARIANT_H256": "4", "FNSSL_NO_STATIC2": "1",
           "FNSSL_LSTM_NO_STATIC": "1", "UNRELATED": "1"}
    e = _lib.tuning_from_env(env)
    assert e.knob[declared["NO_F32_CLUSTER"]] == 1          # getenv()-style switch: set = on, whatever the value
    assert e.knob[declared["CLUSTER_TEST_STALL"]] == 4      # member 3 -> 3 + 1 (0 = no fault injection)
    assert e.knob[declared["LSTM_VARIANT_H256"]] == 4 and e.knob[declared["NO_STATIC2"]] == 1
    assert e.knob[declared["LSTM_NO_STATIC"]] == 1
    with _lib.tuning(bwd_no_cluster=1):
        lib.fnssl_tuning_get(C.byref(back))
        assert back.knob[declared["BWD_NO_CLUSTER"]] == 1
    lib.fnssl_tuning_get(C.byref(back))
    assert back.knob[declared["BWD_NO_CLUSTER"]] == 0


def test_engine_reserves_compute_units_for_rccl_only_when_there_is_an_exchange():
    """TrainEngine._backward_tuning: world 1 -> the process default (None); world > 1 (or reserve_always) -> a per-call
    fnssl_tuning with RESERVED_CUS on top of the current knobs, for the BACKWARD's LSTM calls only (the forward never overlaps
    the gradient all-reduce)."""
    from fnssl import _lib, train
    _lib.load()
    eng = train.TrainEngine.__new__(train.TrainEngine)
    eng.pg, eng.reserved_cus, eng.reserve_always = False, 16, False
    assert eng._backward_tuning() is None
    eng.reserve_always = True
    with _lib.tuning(no_bwd2=1):
        t = eng._backward_tuning()
    names = _lib.tuning_names()
    assert t.knob[names["RESERVED_CUS"]] == 16 and t.knob[names["NO_BWD2"]] == 1
    eng.reserved_cus = 0
    assert eng._backward_tuning() is None
