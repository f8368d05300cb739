"""Multi-GPU equivalence proved on ONE GPU: two processes share cuda:0, talk over the gloo
backend (it accepts device tensors) and run the real HIP path.
  * inference: fnssl.dist.predict_sharded over the HIP predict_step == the single-process output, bit for bit;
  * training: TrainEngine.step on two 1-utterance shards (async per-layer gradient all-reduce, 1/world in Adam)
    leaves the same parameters as one process on the 2-utterance batch — possible because the dropout masks are
    keyed on the GLOBAL pair index, not on the rank.
The 1 -> 8 GPU curve itself can only come from the driver (RCCL over xGMI); this pins the logic it runs."""
import os
import socket
import sys

import numpy as np
import pytest

from conftest import ROOT, assert_close, rs_randn

torch = pytest.importorskip("torch")
pytestmark = pytest.mark.gpu

NT = 24


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _make(dev, seed=31):
    import Model
    from fnssl import weights as W
    sd = W.make_fnssl_state(seed, 4, 256, True)
    net = Model.FN_SSL(is_online=True)
    net.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in sd.items()})
    return net.to(dev)


def _data():
    sig = rs_randn(41, (2, 256 * (NT + 1), 2), 0.1)               # 2 two-mic utterances
    gt = np.tanh(rs_randn(42, (2, NT // 12, 512, 1)))
    return sig, gt


def _worker(rank, world, port, out_dir):
    for p in (os.path.join(ROOT, "fn-ssl_amd"), ROOT, os.path.join(ROOT, "tests")):
        if p not in sys.path:
            sys.path.insert(0, p)
    import torch.distributed as dist
    import predict_step as ps
    from fnssl import dist as fd
    from fnssl import ops, train
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    dev = torch.device("cuda:0")
    torch.cuda.set_device(dev)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    sig, gt = _data()
    # ---- inference: sharded predict, gathered on every rank
    model = ps.MyModel(ch_mode="MM", device=str(dev))
    model.arch = _make(dev)
    model = model.to(dev).eval()
    batch = torch.from_numpy(sig).to(dev).permute(0, 2, 1).contiguous()      # [nb, nch, ns]
    got = fd.predict_sharded(lambda shard: model.predict_step(shard, 0), batch)
    np.save(os.path.join(out_dir, "pred_rank%d.npy" % rank), got.cpu().numpy())
    # ---- training: one step on this rank's utterance
    net = _make(dev)
    eng = train.TrainEngine(net, seed=3)
    lo, hi = train.shard_utterances(2, rank, world)
    x = ops.preprocess(torch.from_numpy(sig[lo:hi]).to(dev), "MM", layout=1)
    loss = eng.step(x, torch.from_numpy(gt[lo:hi]).to(dev))
    np.save(os.path.join(out_dir, "theta_rank%d.npy" % rank), eng.theta.cpu().numpy())
    np.save(os.path.join(out_dir, "grad_rank%d.npy" % rank), eng.grad.cpu().numpy())
    np.save(os.path.join(out_dir, "loss_rank%d.npy" % rank), np.array([loss, eng.last_comm_wait_ms]))
    dist.destroy_process_group()


def test_two_processes_on_one_gpu_match_single_process(tmp_path):
    if not torch.cuda.is_available():
        pytest.fail("-m gpu tests need a ROCm device")
    import torch.multiprocessing as mp
    import predict_step as ps
    from fnssl import ops, train
    world = 2
    mp.spawn(_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    dev = torch.device("cuda:0")
    sig, gt = _data()
    # single-process references through the same HIP path
    model = ps.MyModel(ch_mode="MM", device=str(dev))
    model.arch = _make(dev)
    model = model.to(dev).eval()
    want = model.predict_step(torch.from_numpy(sig).to(dev).permute(0, 2, 1).contiguous(), 0).cpu().numpy()
    for r in range(world):
        np.testing.assert_array_equal(np.load(os.path.join(str(tmp_path), "pred_rank%d.npy" % r)), want)
    net = _make(dev)
    eng = train.TrainEngine(net, seed=3, process_group=False)
    loss = eng.step(ops.preprocess(torch.from_numpy(sig).to(dev), "MM", layout=1), torch.from_numpy(gt).to(dev))
    theta = eng.theta.cpu().numpy()
    th = [np.load(os.path.join(str(tmp_path), "theta_rank%d.npy" % r)) for r in range(world)]
    np.testing.assert_array_equal(th[0], th[1])                   # replicas stay in lockstep
    # all-reduced sum of the two shard gradients / world == the gradient of one pass over both utterances, up to
    # fp32 summation order (the masks are the same because they are keyed on the global pair index)
    g2 = np.load(os.path.join(str(tmp_path), "grad_rank0.npy")) / world
    g1 = eng.grad.cpu().numpy()
    scale = np.abs(g1).max()
    assert np.abs(g2 - g1).max() <= 2e-5 * scale, (np.abs(g2 - g1).max(), scale)
    # Adam's first step is lr * g / (|g| + eps): identical wherever |g| is not down at eps
    big = np.abs(g1) > 1e-4 * scale
    assert_close(th[0][big], theta[big], 0, 2e-6, "parameters after one step (2 ranks vs 1)")
    assert np.abs(th[0] - theta).mean() < 1e-6
    sd0 = _make(torch.device("cpu"))
    moved = np.abs(theta[1:] - np.concatenate([p.detach().numpy().ravel() for p in sd0.parameters()]))
    assert moved.max() > 5e-4                                     # the step did move the parameters
    losses = [np.load(os.path.join(str(tmp_path), "loss_rank%d.npy" % r))[0] for r in range(world)]
    assert abs(0.5 * (losses[0] + losses[1]) - loss) < 1e-5 * max(1.0, abs(loss))   # mean of shard losses = batch loss


def _ddp_worker(rank, world, port, out_dir):
    for p in (os.path.join(ROOT, "fn-ssl_amd"), ROOT, os.path.join(ROOT, "tests")):
        if p not in sys.path:
            sys.path.insert(0, p)
    import torch.distributed as dist
    from torch.nn.parallel import DistributedDataParallel as DDP
    import predict_step as ps
    from fnssl import ops, train
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    dev = torch.device("cuda:0")
    torch.cuda.set_device(dev)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    sig, gt = _data()
    net = _make(dev).train()
    net.dropout_seed = 3                                           # pair_offset None: rank * pairs (equal shards)
    ddp = DDP(net, device_ids=[0])
    opt = torch.optim.Adam(ddp.parameters(), lr=1e-3)
    lo, hi = train.shard_utterances(2, rank, world)
    x = ops.preprocess(torch.from_numpy(sig[lo:hi]).to(dev), "MM", layout=1)
    opt.zero_grad()
    loss = ps._MSELoss.apply(ddp(x), torch.from_numpy(gt[lo:hi]).to(dev))    # main.py:153-154 under strategy="ddp"
    loss.backward()                                                # DDP's reducer all-reduces (averages) the .grads
    flat = torch.cat([p.grad.reshape(-1) for p in net.parameters()])
    np.save(os.path.join(out_dir, "ddp_grad_rank%d.npy" % rank), flat.cpu().numpy())
    opt.step()
    np.save(os.path.join(out_dir, "ddp_theta_rank%d.npy" % rank),
            torch.cat([p.detach().reshape(-1) for p in net.parameters()]).cpu().numpy())
    np.save(os.path.join(out_dir, "ddp_loss_rank%d.npy" % rank), np.array([float(loss)]))
    dist.destroy_process_group()


def test_ddp_over_the_autograd_route_matches_single_process(tmp_path):
    """The reference's multi-GPU mechanism as written (Lightning strategy="ddp", main.py:286-288): DistributedDataParallel
    around the drop-in FN_SSL, train-mode forward with a grad_fn (fnssl/autograd.py), loss.backward() — DDP's own bucket
    all-reduce averages the gradients.  Two ranks x one utterance each == one process on both utterances: same averaged
    gradient (2e-5 of the largest entry), same parameters after a torch.optim.Adam step."""
    if not torch.cuda.is_available():
        pytest.fail("-m gpu tests need a ROCm device")
    import torch.multiprocessing as mp
    import predict_step as ps
    from fnssl import ops
    world = 2
    mp.spawn(_ddp_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    dev = torch.device("cuda:0")
    sig, gt = _data()
    net = _make(dev).train()
    net.dropout_seed = 3
    opt = torch.optim.Adam(net.parameters(), lr=1e-3)
    loss = ps._MSELoss.apply(net(ops.preprocess(torch.from_numpy(sig).to(dev), "MM", layout=1)), torch.from_numpy(gt).to(dev))
    loss.backward()
    g1 = torch.cat([p.grad.reshape(-1) for p in net.parameters()]).cpu().numpy()
    opt.step()
    theta = torch.cat([p.detach().reshape(-1) for p in net.parameters()]).cpu().numpy()
    g = [np.load(os.path.join(str(tmp_path), "ddp_grad_rank%d.npy" % r)) for r in range(world)]
    th = [np.load(os.path.join(str(tmp_path), "ddp_theta_rank%d.npy" % r)) for r in range(world)]
    np.testing.assert_array_equal(g[0], g[1])                      # DDP leaves the same averaged gradient on every rank
    np.testing.assert_array_equal(th[0], th[1])
    scale = np.abs(g1).max()
    assert np.abs(g[0] - g1).max() <= 2e-5 * scale, (np.abs(g[0] - g1).max(), scale)
    big = np.abs(g1) > 1e-4 * scale
    assert_close(th[0][big], theta[big], 0, 2e-6, "parameters after one DDP step (2 ranks vs 1)")
    losses = [np.load(os.path.join(str(tmp_path), "ddp_loss_rank%d.npy" % r))[0] for r in range(world)]
    assert abs(0.5 * (losses[0] + losses[1]) - float(loss)) < 1e-5 * max(1.0, abs(float(loss)))


def _rccl_worker(rank, world, port, out_dir):
    for p in (os.path.join(ROOT, "fn-ssl_amd"), ROOT, os.path.join(ROOT, "tests")):
        if p not in sys.path:
            sys.path.insert(0, p)
    import torch.distributed as dist
    from torch.nn.parallel import DistributedDataParallel as DDP
    import predict_step as ps
    from fnssl import ops, train
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    dev = torch.device("cuda:0")
    torch.cuda.set_device(dev)
    dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)      # "nccl" IS RCCL on ROCm
    sig, gt = _data()
    x = ops.preprocess(torch.from_numpy(sig).to(dev), "MM", layout=1)
    g = torch.from_numpy(gt).to(dev)
    # the fused engine: per-layer gradient all-reduce on the side stream, through RCCL
    eng = train.TrainEngine(_make(dev), seed=3)
    eng.reduce_single_rank = True                                   # the per-bucket asynchronous all-reduces, as with N ranks
    loss = eng.step(x, g)
    assert eng.last_comm_launches == 7, eng.last_comm_launches      # six LSTM layers + the head
    np.save(os.path.join(out_dir, "rccl_engine_theta.npy"), eng.theta.cpu().numpy())
    np.save(os.path.join(out_dir, "rccl_engine_loss.npy"), np.array([loss, float(eng.cluster_fallbacks())]))
    # the reference's route: DDP's reducer over the autograd functions, through RCCL
    net = _make(dev).train()
    net.dropout_seed = 3
    ddp = DDP(net, device_ids=[0])
    opt = torch.optim.Adam(ddp.parameters(), lr=1e-3)
    opt.zero_grad()
    loss = ps._MSELoss.apply(ddp(x), g)
    loss.backward()
    opt.step()
    torch.cuda.synchronize()
    np.save(os.path.join(out_dir, "rccl_ddp_theta.npy"),
            torch.cat([p.detach().reshape(-1) for p in net.parameters()]).cpu().numpy())
    np.save(os.path.join(out_dir, "rccl_ddp_loss.npy"), np.array([float(loss)]))
    dist.destroy_process_group()


def test_one_rank_rccl_group_runs_both_training_routes(tmp_path):
    """What ONE GPU can show of the RCCL path (the N > 1 curve is the driver's): a process group on the "nccl" backend
    (= RCCL) with this library loaded; TrainEngine.step's gradient all-reduce and DistributedDataParallel's reducer over the
    autograd route both go through it, beside the cluster-resident kernels, and leave exactly the parameters of the same
    step without a process group (a one-rank sum / average is the identity)."""
    if not torch.cuda.is_available():
        pytest.fail("-m gpu tests need a ROCm device")
    import torch.multiprocessing as mp
    import predict_step as ps
    from fnssl import ops, train
    mp.spawn(_rccl_worker, args=(1, _free_port(), str(tmp_path)), nprocs=1, join=True)
    dev = torch.device("cuda:0")
    sig, gt = _data()
    x = ops.preprocess(torch.from_numpy(sig).to(dev), "MM", layout=1)
    g = torch.from_numpy(gt).to(dev)
    eng = train.TrainEngine(_make(dev), seed=3)
    loss = eng.step(x, g)
    got = np.load(os.path.join(str(tmp_path), "rccl_engine_loss.npy"))
    assert got[0] == loss and got[1] == 0.0                         # same loss, no guarded fallback beside RCCL
    np.testing.assert_array_equal(np.load(os.path.join(str(tmp_path), "rccl_engine_theta.npy")), eng.theta.cpu().numpy())
    net = _make(dev).train()
    net.dropout_seed = 3
    opt = torch.optim.Adam(net.parameters(), lr=1e-3)
    loss = ps._MSELoss.apply(net(x), g)
    loss.backward()
    opt.step()
    assert np.load(os.path.join(str(tmp_path), "rccl_ddp_loss.npy"))[0] == float(loss)
    np.testing.assert_array_equal(np.load(os.path.join(str(tmp_path), "rccl_ddp_theta.npy")),
                                  torch.cat([p.detach().reshape(-1) for p in net.parameters()]).cpu().numpy())


def test_bench_self_launches_n_ranks(tmp_path):
    """`python bench.py --gpus 2` with no WORLD_SIZE must itself start 2 ranks (torch.distributed.run, one process per
    GPU).  This box has ONE GPU, so (a) without the test hook the call must exit non-zero, not fall back to one rank;
    (b) with FNSSL_BENCH_SHARED_GPU=1 both ranks share cuda:0 over gloo (RCCL refuses two ranks on one device) and the
    whole launch -> barrier -> max-over-ranks -> ONE JSON line path runs, for the forward and the training step."""
    import json
    import subprocess
    if not torch.cuda.is_available():
        pytest.fail("-m gpu tests need a ROCm device")
    env = dict(os.environ)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "FNSSL_BENCH_SHARED_GPU"):
        env.pop(k, None)
    bench = os.path.join(ROOT, "bench.py")
    if torch.cuda.device_count() < 2:
        res = subprocess.run([sys.executable, bench, "--gpus", "2", "--steps", "1", "--warmup", "1"], env=env,
                             stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=600)
        assert res.returncode != 0 and res.stdout.strip() == "", (res.returncode, res.stdout[-300:])
    env["FNSSL_BENCH_SHARED_GPU"] = "1"
    for cfg, extra in ((2, ["--nb", "2", "--frames", "24"]), (4, ["--nb", "2", "--frames", "24"])):
        res = subprocess.run([sys.executable, bench, "--gpus", "2", "--config", str(cfg), "--steps", "2", "--warmup", "1",
                              "--other-configs", ""] + extra, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                             text=True, timeout=900)
        assert res.returncode == 0, res.stderr[-2000:]
        lines = [ln for ln in res.stdout.splitlines() if ln.strip()]
        assert len(lines) == 1, res.stdout[-500:]
        assert len(lines[0]) < 4096, "the stdout line must stay compact (the driver keeps ~10 KB): %d bytes" % len(lines[0])
        line = json.loads(lines[0])
        assert line["roofline"] is None or "frac" in line["roofline"]
        assert os.path.exists(os.path.join(ROOT, line["detail"])), "the full record was not written"
        assert line["n_gpus"] == 2 and len(line["ms_per_step_per_rank"]) == 2 and line["backend"] == "gloo"
        assert line["value"] > 0 and line["scaling"] == "weak"
        # whole-job value = frames of BOTH ranks over the slowest rank's time
        assert abs(line["value"] - 2 * 2 * 24 / (line["ms_per_step"] * 1e-3)) <= 1e-3 * line["value"] + 0.01
    # strong scaling: the configuration's batch is the GLOBAL batch, split over the ranks; the line carries its
    # own efficiency against one rank running the whole batch in the same run
    res = subprocess.run([sys.executable, bench, "--gpus", "2", "--config", "2", "--scaling", "strong", "--steps", "2", "--warmup", "1",
                          "--nb", "2", "--frames", "24"], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=900)
    assert res.returncode == 0, res.stderr[-2000:]
    line = json.loads([ln for ln in res.stdout.splitlines() if ln.strip()][0])
    assert line["scaling"] == "strong" and line["n_gpus"] == 2 and "other_configs" not in line
    assert line["config"]["utterances_per_gpu"] == 1 and line["config"]["global_batch"] == 2
    assert abs(line["value"] - 2 * 24 / (line["ms_per_step"] * 1e-3)) <= 1e-3 * line["value"] + 0.01
    ss = line["strong_scaling"]
    assert ss["global_batch"] == 2 and ss["utterances_per_rank"] == 1 and 0.0 < ss["efficiency"] < 4.0
