"""W > 1 execution of the PRODUCT's data-parallel path: W processes, each running taper_amd's Trainer on its
shard of every global batch with a Communicator between backward and Adam::step.  Checked: replicas bit-identical, and
weights / losses equal to ONE process training on the full global batch (the mean of shard-mean gradients is the full-batch
gradient).  The RCCL communicator needs one GPU per rank (skipped on a 1-GPU box); the peer-to-peer communicator
(th_p2p_*: each rank reads its peers' gradient arenas through IPC handles) also runs with every rank on ONE GPU, which
is how the 1-GPU CI box executes a real W = 2 / W = 4 step."""
import os
import subprocess
import sys
import uuid
from pathlib import Path

import numpy as np
import pytest

from tests import margins

pytestmark = pytest.mark.gpu
BOUND_DP_LOSS = 1.3e-6   # mean of the shard losses vs one process / the oracle: observed <= 6.0e-7 (2x the r04 observation, profiles/r04_parity_margins.json)
BOUND_DP_LR = 8.4e-3   # weights after 2 x steps Adam steps vs one process / the oracle, in units of lr: observed <= 4.17e-3 (2x the r04 observation, profiles/r04_parity_margins.json)
ROOT = Path(__file__).resolve().parent.parent


def _run_ranks(tmp_path, world, backend, mode, steps, global_batch, same_device, fuse=True, model="mlp_baseline", fine=False, inkernel=True):
    env0 = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    env0["TAPER_DP_MODEL"] = model
    env0["TAPER_DP_INKERNEL"] = "1" if inkernel else "0"
    env0["TAPER_DP_FINE"] = "1" if fine else "0"
    env0["TAPER_P2P_FUSE"] = "1" if fuse else "0"
    key = uuid.uuid4().hex[:12]
    procs = []
    for r in range(world):
        env = dict(env0, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT="29561",
                   TAPER_DP_OUT=str(tmp_path), TAPER_DP_STEPS=str(steps), TAPER_DP_GLOBAL_BATCH=str(global_batch), TAPER_DP_MODE=mode,
                   TAPER_DP_BACKEND=backend, TAPER_DP_KEY=key, HSA_ENABLE_IPC_MODE_LEGACY="0")
        if same_device:
            env["TAPER_DP_DEVICE"] = "0"
        procs.append(subprocess.Popen([sys.executable, str(ROOT / "tests" / "dp_worker.py")], env=env, stdout=subprocess.PIPE,
                                      stderr=subprocess.STDOUT, text=True))
    outs = []
    for p in procs:
        try:
            o, _ = p.communicate(timeout=240)
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise
        outs.append(o)
    for r, p in enumerate(procs):
        assert p.returncode == 0, f"rank {r} failed:\n{outs[r][-3000:]}"
    return [np.load(tmp_path / f"rank{r}.npz") for r in range(world)]


def _single_process_reference(steps, global_batch, model_name="mlp_baseline"):
    """the same optimisation in ONE process on the full global batches (two epochs, like the workers)"""
    import taper_amd as T
    from tests import backends
    from tests.dp_worker import make_problem, sample_shape
    spec, x, y = make_problem(steps, global_batch, model=model_name)
    H = backends.get("hip")
    model = H.sequential(spec)
    opt = T.Adam(model.parameters(), 1e-3, None, None, 1e-4)
    tr = T.Trainer(model, opt, **({"sample_shape": sample_shape(model_name)} if sample_shape(model_name) else {}))
    loader = T.DataLoader(T.MNISTDataset.from_host(x, y), global_batch, False)
    losses = np.concatenate([tr.run_epoch(loader, T.Trainer.GRAPH)["losses"] for _ in range(2)])
    return losses, [p.data() for p in model.parameters()], opt.t()


def _check(ranks, world, steps, global_batch, model="mlp_baseline"):
    ref_losses, ref_params, ref_t = _single_process_reference(steps, global_batch, model)
    n_params = len(ref_params)
    for r in range(world):
        assert int(ranks[r]["t"]) == ref_t == 2 * steps
        for i in range(n_params):
            np.testing.assert_array_equal(ranks[r][f"p{i}"], ranks[0][f"p{i}"], err_msg=f"rank {r} param {i}: replicas diverged")
    # loss of the global batch = mean of the shard losses (each a mean over B/W rows)
    mean_losses = np.mean([ranks[r]["losses"] for r in range(world)], axis=0)
    margins.check("mean_losses_vs_one_process", mean_losses, ref_losses, BOUND_DP_LOSS)
    for i in range(n_params):
        margins.check(f"param{i}_vs_one_process", ranks[0][f"p{i}"], ref_params[i], BOUND_DP_LR, lr=1e-3)


@pytest.mark.parametrize("mode", ["graph", "eager"])
def test_rccl_two_ranks_equal_full_batch(tmp_path, mode):
    from taper_amd import hip
    if hip.device_count() < 2:
        pytest.skip("RCCL needs one GPU per rank; this box has %d" % hip.device_count())
    ranks = _run_ranks(tmp_path, 2, "rccl", mode, steps=6, global_batch=256, same_device=False)
    _check(ranks, 2, 6, 256)


@pytest.mark.parametrize("world,mode,fuse", [(2, "graph", True), (2, "eager", True), (4, "graph", True), (2, "graph", False), (8, "graph", True),
                                             (8, "eager", False)])
def test_p2p_ranks_on_one_gpu_equal_full_batch(tmp_path, world, mode, fuse):
    """the THREE-launch step (gradient launch, then the one-shot peer-to-peer all-reduce) with every rank on GPU 0: a real multi-process step
    on the 1-GPU box.  fuse: all-reduce + Adam in one launch (th_allreduce_adam); otherwise th_allreduce_sum_scale in place, then Adam::step"""
    ranks = _run_ranks(tmp_path, world, "p2p", mode, steps=6, global_batch=256, same_device=True, fuse=fuse, inkernel=False)
    _check(ranks, world, 6, 256)
    for r in ranks:     # the path under test is the one that ran (eager: one launch per step; graph: one per captured step + the eager ones)
        fused, inplace = int(r["launches_fused"]), int(r["launches_inplace"])
        assert int(r["launches_inkernel"]) == 0
        if fuse:     # the bootstrap's self-check: 3 rounds through each kernel
            assert fused >= 6 + 3 and inplace == 3, (fused, inplace)
        else:
            assert inplace >= 6 + 3 and fused == 0, (fused, inplace)


@pytest.mark.parametrize("mode,global_batch", [("graph", 256), ("graph", 128), ("graph", 64)])
def test_p2p_two_ranks_exchange_inside_the_gradient_launch(tmp_path, mode, global_batch):
    """the TWO-launch data-parallel step (th_mlp_tail_dp): every workgroup of the gradient launch pushes its finished slice to the peer,
    takes the peer's, and applies Adam to the mean from registers -- two ranks on GPU 0 (the wait graph has no cycle with one peer on the
    device: th_mlp_tail_dp_supported).  Replicas bit-identical; weights and losses equal to one process on the full batches and to the
    oracle's loop; no all-reduce launch ran beyond the bootstrap's self-check."""
    ranks = _run_ranks(tmp_path, 2, "p2p", mode, steps=6, global_batch=global_batch, same_device=True)
    _check(ranks, 2, 6, global_batch)
    _check_against_oracle(ranks, 2, 6, global_batch)
    for r in ranks:
        assert int(r["launches_inkernel"]) >= 6, int(r["launches_inkernel"])
        assert int(r["launches_fused"]) == 3 and int(r["launches_inplace"]) == 3, (int(r["launches_fused"]), int(r["launches_inplace"]))


def test_p2p_two_ranks_two_shot_exchange(tmp_path, monkeypatch):
    """the TWO-shot form of the in-launch exchange (csrc/dp_dev.h: slice s is reduced by rank s % W, the mean pushed back -- what rings of
    four ranks and more take, 2 / W of the bytes per link), forced on two ranks so that whole training runs go through it on the 1-GPU box:
    the same bits as the one-shot form gives (both add in rank order), replicas identical, equal to one process and to the oracle"""
    monkeypatch.setenv("TAPER_DP_TWO_SHOT", "1")
    two = _run_ranks(tmp_path / "two", 2, "p2p", "graph", steps=6, global_batch=256, same_device=True) if (tmp_path / "two").mkdir() is None else None
    _check(two, 2, 6, 256)
    _check_against_oracle(two, 2, 6, 256)
    assert all(int(r["exchange_form"]) == 2 and int(r["launches_inkernel"]) >= 6 for r in two)
    monkeypatch.setenv("TAPER_DP_TWO_SHOT", "0")
    one = _run_ranks(tmp_path / "one", 2, "p2p", "graph", steps=6, global_batch=256, same_device=True) if (tmp_path / "one").mkdir() is None else None
    assert all(int(r["exchange_form"]) == 1 for r in one)
    for i in range(4):
        np.testing.assert_array_equal(two[0][f"p{i}"], one[0][f"p{i}"], err_msg=f"param {i}: the two forms differ")
    np.testing.assert_array_equal(two[0]["losses"], one[0]["losses"])


def test_p2p_four_ranks_two_shot_exchange_inside_the_gradient_launch(tmp_path, monkeypatch):
    """FOUR processes on GPU 0 training through the in-launch exchange in its two-shot form (what four ranks and more take by default): the
    784-64-10 model's tail launch is 108 workgroups of four waves at 64 rows per rank, so three ranks' worth of waiting workgroups leave
    places free on one device.  The default rule for ranks that share a device also keeps a CU free in every shader engine, counting each
    waiting workgroup as a whole CU (comm_dp_shared_fits, csrc/comm.hip: three ranks of 128 rows starved one another without it,
    profiles/r06_dp_three_ranks_one_device.txt), which refuses this configuration; its four-wave waiting workgroups do share CUs with the
    first launch's (placement trace: 8 - 13 of 16 tiles land beside a waiting workgroup), so this PROTOCOL test asks for the places rule
    alone -- 48 of 48 runs in r06.  Replicas bit-identical; weights and losses equal to one process on the 256-row batches and to the
    oracle's loop."""
    monkeypatch.setenv("TAPER_DP_SHARED_RULE", "places")
    ranks = _run_ranks(tmp_path, 4, "p2p", "graph", steps=6, global_batch=256, same_device=True, model="mlp_64")
    _check(ranks, 4, 6, 256, model="mlp_64")
    _check_against_oracle(ranks, 4, 6, 256, model="mlp_64")
    for r in ranks:
        assert int(r["exchange_form"]) == 2 and int(r["launches_inkernel"]) >= 6, (int(r["exchange_form"]), int(r["launches_inkernel"]))
        assert int(r["launches_fused"]) == 3          # the bootstrap's self-check only


@pytest.mark.parametrize("world", [2, 3, 8])
def test_p2p_exchange_alone_many_rounds(tmp_path, world, monkeypatch):
    """the exchange on its own (th_comm_exchange_selftest: 16 workgroups per rank, known patterns checked on the device) for 1 500 rounds
    through the same words, W processes on GPU 0 -- the one-shot form at W = 2 / 3, the two-shot form at W = 8 (owners 0 .. 7 all in play)"""
    monkeypatch.setenv("TAPER_DP_SELFTEST_ROUNDS", "1500")
    ranks = _run_ranks(tmp_path, world, "p2p", "graph", steps=2, global_batch=64 * world, same_device=True)
    for r in ranks:
        assert int(r["selftest_bad"]) == 0 and int(r["exchange_form"]) == (2 if world >= 4 else 1)


@pytest.mark.parametrize("world,model", [(4, "mlp_baseline"), (8, "mlp_baseline"), (3, "mlp_64"), (3, "cnn_simple")])
def test_p2p_many_ranks_on_one_gpu_fall_back_to_the_three_launch_step(tmp_path, world, model):
    """a workgroup that waits for a peer's slice holds its place on the device, and the late rank still has to get the launches in front
    of its exchange launch dispatched -- whole-CU workgroups, dealt to the shader engines in rotation: the waiting workgroups of the other
    ranks must fit the device's places AND leave a CU free in every engine (comm_dp_shared_fits, csrc/comm.hip).  Four or eight ranks of
    the 784-128-10 model fail the first condition; three ranks of the 784-64-10 model or of the simple CNN the second (2 x ceil(108 / 32)
    = 8 > 7: they starved the late rank's first launch in 10 of 20 runs, profiles/r06_dp_three_ranks_one_device.txt) -- the Trainer takes
    the three-launch step (on a node with a GPU per rank every grid is resident on its own device and the answer is always yes)"""
    steps, gb = (4, 64 * world) if model == "mlp_baseline" else (3, 128 * world)
    ranks = _run_ranks(tmp_path, world, "p2p", "graph", steps=steps, global_batch=gb, same_device=True, model=model)
    _check(ranks, world, steps, gb, model=model)
    for r in ranks:
        assert int(r["launches_inkernel"]) == 0 and int(r["launches_fused"]) >= steps + 3
        # (the bootstrap's self-check still ran the exchange on its own, 16 workgroups per rank, in the form that many ranks take)
        assert int(r["exchange_form"]) == (2 if world >= 4 else 1)


@pytest.mark.parametrize("world,global_batch", [(2, 256), (2, 512)])
def test_p2p_simple_cnn_exchange_inside_the_batch_sums_launch(tmp_path, world, global_batch):
    """BASELINE configs[2]'s model data-parallel: the simple CNN's step keeps its TWO launches -- the chain with the classifier's rows, then the
    batch sums (th_wide_head_grads_dp), whose workgroups exchange every finished sum with the peers (dW blocks, db, the last conv's bias) and
    apply Adam to the mean.  No all-reduce launch beyond the bootstrap's self-check; replicas bit-identical; weights and losses equal to one
    process on the global batches.  (Two ranks on GPU 0: 103 workgroups of 16 waves per rank, one to a CU -- the peer's chain launch, a CU per
    workgroup too, runs on the other 153.)"""
    ranks = _run_ranks(tmp_path, world, "p2p", "graph", steps=3, global_batch=global_batch, same_device=True, model="cnn_simple")
    _check(ranks, world, 3, global_batch, model="cnn_simple")
    for r in ranks:
        assert int(r["launches_inkernel"]) >= 3, int(r["launches_inkernel"])
        assert int(r["launches_fused"]) == 3 and int(r["exchange_form"]) == (2 if world >= 4 else 1)


def test_p2p_reference_cnn_two_ranks_equal_full_batch(tmp_path):
    """the reference CNN (examples/train_mnist_cnn.rs) data-parallel, 128 rows per rank: the conv chain and th_mlp3_xent write plain gradients
    (no Adam in their epilogues: the all-reduce comes first), the one-shot all-reduce + Adam finishes the step; replicas bit-identical and
    equal to the single-process step on the 256-row batches"""
    ranks = _run_ranks(tmp_path, 2, "p2p", "graph", steps=3, global_batch=256, same_device=True, model="cnn_reference")
    _check(ranks, 2, 3, 256, model="cnn_reference")


def _oracle_reference(steps, global_batch, model_name="mlp_baseline"):
    """the ORACLE's training loop on the full global batches (two epochs, like the workers): /root/reference/src/train.rs:98-144 over
    examples/train_mnist.rs's model, restated in oracle/ -- the data-parallel run must land where the reference's single process lands"""
    from tests import backends
    from tests.dp_worker import make_problem, sample_shape
    spec, x, y = make_problem(steps, global_batch, model=model_name)
    Orc = backends.get("oracle")
    Orc.set_zero_sentinel(True)
    om = Orc.sequential(spec)
    oopt = Orc.m.Adam(om.parameters(), 1e-3, None, None, 1e-4)
    shape = (global_batch,) + (sample_shape(model_name) or (784,))
    losses = []
    for _ in range(2):
        for s_ in range(steps):
            r = om.train_step(oopt, x[s_ * global_batch:(s_ + 1) * global_batch], y[s_ * global_batch:(s_ + 1) * global_batch], shape)
            losses.append(r["loss"])
    return np.asarray(losses), [p.data() for p in om.parameters()]


def _check_against_oracle(ranks, world, steps, global_batch, model="mlp_baseline"):
    o_losses, o_params = _oracle_reference(steps, global_batch, model)
    mean_losses = np.mean([ranks[r]["losses"] for r in range(world)], axis=0)
    margins.check("mean_losses_vs_oracle", mean_losses, o_losses, BOUND_DP_LOSS)
    for i, op in enumerate(o_params):
        margins.check(f"param{i}_vs_oracle", ranks[0][f"p{i}"], op, BOUND_DP_LR, lr=1e-3)


@pytest.mark.parametrize("fuse", [True, False])
def test_p2p_eight_ranks_of_128_rows_is_baseline_configs_3(tmp_path, fuse):
    """BASELINE configs[3] at its stated shape: the MLP, global batch 1024 = 8 ranks x 128 rows, gradient all-reduce between backward and
    Adam -- here with the ranks sharing GPU 0 (the box has one), over the peer-to-peer communicator.  Replicas bit-identical; rank 0's
    weights and the global losses equal to one HIP process on the 1024-row batches AND to the oracle's training loop on them."""
    ranks = _run_ranks(tmp_path, 8, "p2p", "graph", steps=4, global_batch=1024, same_device=True, fuse=fuse)
    assert all(len(r["losses"]) == 8 for r in ranks)
    _check(ranks, 8, 4, 1024)
    _check_against_oracle(ranks, 8, 4, 1024)


def test_p2p_two_ranks_of_512_rows_take_the_three_launch_step(tmp_path):
    """shards large enough for th_mlp2_xent (>= 480 rows): its finish launch writes plain gradients into the optimizer's arena (no Adam in
    its epilogue: the all-reduce comes first), the one-shot all-reduce + Adam finishes the step.  Replicas bit-identical; weights and losses
    equal to one process on the 1024-row batches (itself on th_mlp2_xent) and to the oracle's loop"""
    ranks = _run_ranks(tmp_path, 2, "p2p", "graph", steps=4, global_batch=1024, same_device=True)
    assert all(len(r["losses"]) == 8 for r in ranks)
    assert all(int(r["mlp2_calls"]) > 0 for r in ranks), "the shards did not take th_mlp2_xent"
    _check(ranks, 2, 4, 1024)
    _check_against_oracle(ranks, 2, 4, 1024)


def test_p2p_two_ranks_against_the_oracle(tmp_path):
    ranks = _run_ranks(tmp_path, 2, "p2p", "graph", steps=6, global_batch=256, same_device=True)
    _check_against_oracle(ranks, 2, 6, 256)


@pytest.mark.parametrize("world", [2, 8])
def test_p2p_fine_grained_arena(tmp_path, world):
    """the gradient arena in fine-grained device memory (th_malloc_finegrained: never cached in a peer's L2) -- the fallback `auto` takes
    when the multi-round self-check fails on the pooled arena: same results"""
    ranks = _run_ranks(tmp_path, world, "p2p", "graph", steps=6, global_batch=256, same_device=True, fine=True)
    _check(ranks, world, 6, 256)
    assert all(int(r["fine"]) == 1 for r in ranks)


def test_p2p_ranks_on_separate_gpus(tmp_path):
    from taper_amd import hip
    if hip.device_count() < 2:
        pytest.skip("needs 2 GPUs; this box has %d" % hip.device_count())
    ranks = _run_ranks(tmp_path, 2, "p2p", "graph", steps=6, global_batch=256, same_device=False)
    _check(ranks, 2, 6, 256)


def test_bench_self_spawn_two_ranks_on_one_gpu():
    """`python bench.py --gpus 2` with no launcher spawns the ranks itself; with the test hook that lets them share GPU 0 the whole
    N > 1 bench flow (rendezvous, p2p bootstrap + self-check, timed region with barriers, replica check, single-GPU figure of the
    same per-GPU batch) runs on the 1-GPU box"""
    import json
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    env.update(TAPER_BENCH_SHARE_DEVICE="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--gpus", "2", "--steps", "40", "--warmup", "5", "--dp-backend", "p2p"], env=env,
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]                  # rank 0 prints ONE line
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 40 and d["config"]["workload"] == "mlp_784-128-10_b128"   # BASELINE configs[3]: 128 rows per GPU
    assert d["config"]["global_batch"] == 256 and d["config"]["parallelism"] == "dp2" and "p2p" in d["config"]["comm"]
    assert len(lines[0]) < 2000                               # the whole line fits the driver's 2 000-character tail
    dp = d["data_parallel"]
    assert dp["replicas_bit_identical"] is True
    assert dp["single_gpu_ms_per_step"] > 0 and dp["b64_per_gpu_ms_per_step"] > 0
    # eff = T(1 GPU, B/W rows) / T(W GPUs, B/W rows each) -- but ranks that SHARE a device time-share it: null, not an artefact
    assert dp["weak_scaling_efficiency"] is None
    assert "skipped" in dp["same_job_over_rccl"]              # two ranks on one device: RCCL cannot run, and the line says so
    rf = d["roofline"]                                        # N > 1, the exchange inside the gradient launch: the whole two-launch step
    assert rf["kernel"].startswith("sgemm_small16_tick + mlp_tail_exact_kernel") and rf["bound"] == "hbm" and rf["us_per_launch"] > 0
    assert rf["us_per_launch"] == pytest.approx(d["ms_per_step"] * 1e3, rel=1e-2)
    assert rf["frac"] == pytest.approx(rf["achieved"] / rf["peak"], rel=1e-3) and "HBM" in rf["peak_basis"]
    P = 101772                                                # 100 352 + 128 + 1 280 + 12 (slices padded to 4 floats)
    assert rf["alg_bytes_per_launch"] > 2 * (2 - 1) * 4 * P   # the step's own bytes + the slices pushed to and read from one peer
    assert d["step_roofline"]["mfma_frac"] > 0
    assert d["cpu_baseline"]["value"] is None and "--gpus 1" in d["cpu_baseline"]["see"]
    assert d["value"] > 0 and d["value"] == pytest.approx(40 * 256 / (d["ms_per_step"] * 1e-3 * 40), rel=1e-3)
    full = json.loads((ROOT / d["details"]).read_text())      # the side file keeps everything the line left out
    assert "inside the gradient launch" in full["data_parallel"]["exchange_form"]
    assert full["data_parallel"]["single_gpu_same_per_gpu_batch"]["per_gpu_batch"] == 128
    assert full["data_parallel"]["dp_at_64_rows_per_gpu"]["replicas_bit_identical"] is True


def test_bench_eight_ranks_is_global_batch_1024():
    """`python bench.py --gpus 8` is BASELINE configs[3]: 128 rows per GPU, global batch 1024 (ranks share GPU 0 through the test hook)"""
    import json
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    env.update(TAPER_BENCH_SHARE_DEVICE="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--gpus", "8", "--steps", "20", "--warmup", "5"], env=env,
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 8 and d["config"]["global_batch"] == 1024 and d["config"]["workload"] == "mlp_784-128-10_b128"
    assert d["config"]["parallelism"] == "dp8" and "p2p" in d["config"]["comm"]          # --dp-backend auto took the one-shot form
    assert d["data_parallel"]["replicas_bit_identical"] is True and d["scaling"] == "weak"
    assert len(lines[0]) < 2000
    # (eight ranks on ONE device: the three-launch step -- th_mlp_tail_dp_supported -- and no efficiency figure from time-shared ranks)
    assert d["data_parallel"]["weak_scaling_efficiency"] is None and d["roofline"]["alg_bytes_per_launch"] == (8 - 1 + 1 + 6) * 4 * 101772
    assert d["roofline"]["kernel"].startswith("p2p_allreduce_adam")


@pytest.mark.parametrize("form", ["fused", "inplace", "inkernel", "inkernel_cnn"])
def test_p2p_missing_peer_times_out_instead_of_hanging(tmp_path, form):
    """a peer that never launches its side of the all-reduce: the waiting rank's kernel gives up after its wall-clock-bounded spin and the
    communicator reports it -- no GPU hang -- and NOTHING was applied, in both forms: the fused all-reduce + Adam launch skips its update,
    and behind the in-place all-reduce (which stores nothing) Adam::step skips itself on the communicator's error word"""
    env0 = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    key = uuid.uuid4().hex[:12]
    procs = [subprocess.Popen([sys.executable, str(ROOT / "tests" / "p2p_straggler_worker.py")],
                              env=dict(env0, RANK=str(r), WORLD_SIZE="2", TAPER_DP_OUT=str(tmp_path), TAPER_DP_KEY=key, HSA_ENABLE_IPC_MODE_LEGACY="0",
                                       TAPER_STRAGGLER_FORM=form),
                              stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True) for r in range(2)]
    outs = []
    for p in procs:
        try:
            o, _ = p.communicate(timeout=120)
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise
        outs.append(o)
    for r, p in enumerate(procs):
        assert p.returncode == 0, f"rank {r} failed:\n{outs[r][-3000:]}"
    flag, seconds, unchanged, t = (tmp_path / "straggler_result.txt").read_text().split()
    assert flag == "1"                       # both steps raised "timed out waiting for a peer"; th_comm_error / _peek agree
    assert 2.0 < float(seconds) < 30.0       # after the 3 s bound (the second step returns at once: the error is final), not never
    assert unchanged == "1" and t == "0"     # the step that timed out applied nothing and did not advance Adam's counter
