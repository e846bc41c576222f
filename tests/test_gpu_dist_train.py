"""Data-parallel training end to end on the GPU box: two processes (both on cuda:0, gloo for the exchange because
RCCL wants one GPU per rank) run GeneratorModel.train_on_batch on their halves of a batch -- sr_train_grads into
the torch gradient buffer, mean over the ranks, sr_train_apply -- and must end with the weights a single
process gets from the whole batch; then one device-resident iteration (train_on_clips) on one clip per rank against
a single process on both clips.  On a multi-GPU node the same code runs with backend nccl (= RCCL)."""
import os
import socket
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _batch(B, h, w):
    rng = np.random.default_rng(77)
    return [rng.random((B, h, w, 3), dtype=np.float32), rng.random((B, h, w, 3), dtype=np.float32),
            rng.random((B, 4 * h, 4 * w, 3), dtype=np.float32), rng.random((B, 4 * h, 4 * w, 3), dtype=np.float32)]


def _clips(n):
    return np.random.default_rng(5).integers(0, 256, (n, 2, 64, 64, 3), dtype=np.uint8)   # 2 frames: fits the max_batch of the steps before


def _worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK="0", SR_DIST_BACKEND="gloo")
    for p in (ROOT, os.path.join(ROOT, "super-resolution_amd")):
        if p not in sys.path:
            sys.path.insert(0, p)
    import dp
    from layers import Adam, GeneratorCore, GeneratorModel
    from oracle import ref_numpy as rn
    core = GeneratorCore((16, 16, 3), dtype="fp32", device=0, seed=100 + rank)      # different random weights per rank
    m = GeneratorModel(core, "FRVSR")
    m.compile(loss=["mean_squared_error"] * 2, loss_weights=[1., 1.], optimizer=Adam(lr=1e-4))
    dp.init_if_needed()
    w0 = list(rn.make_weights(seed=7).items()) if rank == 0 else core.tensors
    core.set_tensors(dp.broadcast_weights_(w0, 0))                                    # rank 0's weights everywhere
    a, b, c, hr = _batch(2 * world, 16, 16)
    sl = slice(2 * rank, 2 * rank + 2)
    losses = []
    for _ in range(2):
        losses.append(m.train_on_batch([a[sl], b[sl], c[sl]], [hr[sl], a[sl]]))
    w_batch = [x.copy() for x in m.get_weights()]
    l_clips = m.train_on_clips(_clips(world)[rank:rank + 1])
    out.put((rank, losses, w_batch, l_clips, [x.copy() for x in m.get_weights()]))
    import torch.distributed as dist
    dist.barrier()
    dist.destroy_process_group()


def test_two_ranks_train_like_one_process_on_the_whole_batch(weights7):
    import torch.multiprocessing as mp
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = sorted([q.get(timeout=240) for _ in range(world)], key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    # one process, whole batch
    from layers import Adam, GeneratorCore, GeneratorModel
    core = GeneratorCore((16, 16, 3), dtype="fp32", device=0)
    core.set_tensors(list(weights7.items()))
    m = GeneratorModel(core, "FRVSR")
    m.compile(loss=["mean_squared_error"] * 2, loss_weights=[1., 1.], optimizer=Adam(lr=1e-4))
    a, b, c, hr = _batch(2 * world, 16, 16)
    want_l = [m.train_on_batch([a, b, c], [hr, a]) for _ in range(2)]
    want_w = [x.copy() for x in m.get_weights()]
    want_lc = m.train_on_clips(_clips(world))
    want_wc = m.get_weights()
    for rank, losses, weights, l_clips, weights_c in got:
        assert np.allclose(losses, want_l, rtol=1e-5), (rank, losses, want_l)          # mean of the shard losses = batch loss
        d = np.concatenate([np.abs(x - y).ravel() for x, y in zip(weights, want_w)])
        # same gradient up to summation order -> same Adam steps (2 x 1e-4); elements with a gradient at rounding
        # level may differ by a fraction of a step
        assert d.mean() <= 0.01 * 2e-4 and np.quantile(d, 0.999) <= 0.5 * 2e-4, (rank, d.mean(), d.max())
        # the device-resident iteration shards by clip: one clip per rank == both clips in one process
        # (the weights the ranks start it from differ from the single process's by the rounding-level steps above)
        assert np.allclose(l_clips, want_lc, rtol=1e-3), (rank, l_clips, want_lc)
        d = np.concatenate([np.abs(x - y).ravel() for x, y in zip(weights_c, want_wc)])
        assert d.mean() <= 0.01 * 3e-4 and np.quantile(d, 0.999) <= 0.5 * 3e-4, (rank, d.mean(), d.max())
    for k in (2, 4):
        for x, y in zip(got[0][k], got[1][k]):
            np.testing.assert_array_equal(x, y)                                        # the ranks stay in lock step exactly


def _rccl_worker(port, out, direct):
    """ONE rank on backend nccl (= RCCL): the gradient goes through the same all-reduce a multi-GPU job runs -- the library's own
    ncclAllReduce on its own communicator (direct), or torch.distributed's (SR_DP_DIRECT=0)."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", WORLD_SIZE="1", LOCAL_RANK="0",
                      SR_DP_FORCE="1", SR_DIST_BACKEND="nccl", SR_DP_DIRECT="1" if direct else "0")
    for p in (ROOT, os.path.join(ROOT, "super-resolution_amd")):
        if p not in sys.path:
            sys.path.insert(0, p)
    import dp
    import torch.distributed as dist
    from layers import Adam, Discriminator, GeneratorCore, GeneratorModel, TecoGANModel
    from oracle import ref_numpy as rn
    assert dp.active()
    core = GeneratorCore((16, 16, 3), model="tecogan", dtype="fp32", device=0)
    core.set_tensors(list(rn.make_weights(seed=7).items()))
    m = GeneratorModel(core, "Generator")
    opt = Adam(lr=1e-4)
    m.compile(loss=["mean_squared_error"] * 2, loss_weights=[1., 1.], optimizer=opt)
    a, b, c, hr = _batch(2, 16, 16)
    losses = m.train_on_batch([a, b, c], [hr, a])
    assert dist.is_initialized() and dist.get_backend() == "nccl" and dist.get_world_size() == 1
    assert dp.direct() == direct and (dp._comm is not None) == direct
    if direct:
        # the communicator on a plain device buffer: a one-rank sum leaves it as it was
        import torch
        t = torch.arange(1000, dtype=torch.float32, device="cuda:0")
        dp._comm.allreduce_f32(t.data_ptr(), t.numel())
        assert dp._comm.nranks == 1 and np.array_equal(t.cpu().numpy(), np.arange(1000, dtype=np.float32))
        # the C-level sequence of INTEGRATION.md on the ctx's OWN gradient buffer: grads -> all-reduce -> apply, no host wait
        # in between; one rank: the gradient and the step are those of the plain sequence
        import sr_hip
        plain = sr_hip.Context(16, 16, max_batch=2, dtype="fp32")
        plain.load_weights(rn.make_weights(seed=7))
        _, g_plain = plain.train_grads(a, b, c, hr, read=True)
        plain.train_apply(lr=1e-4)
        seq = sr_hip.Context(16, 16, max_batch=2, dtype="fp32")
        seq.load_weights(rn.make_weights(seed=7))
        seq.train_grads(a, b, c, hr)
        seq.train_allreduce(dp._comm)
        assert np.array_equal(seq.train_read_grads(), g_plain)
        seq.train_apply(grad_scale=1.0 / dp._comm.nranks, lr=1e-4)
        for (_, x), (_, y) in zip(seq.get_weights(), plain.get_weights()):
            assert np.array_equal(x, y)
        plain.close()
        seq.close()
    w_gen = [x.copy() for x in m.get_weights()]
    # the adversarial models use the same path: discriminator step and combined step through RCCL
    d = Discriminator((64, 64, 3), device=0).build()
    d.set_weights([np.asarray(x) for x in _disc_weights()])
    d.compile(loss=["binary_crossentropy"] + ["mean_squared_error"] * 4, loss_weights=[1., 0., 0., 0., 0.], optimizer=opt)
    imgs = _gan_images()
    dl = d.train_on_batch(imgs, [np.ones(2, np.float32)] + [None] * 4)
    feats = d.predict(imgs)[1:]
    g = TecoGANModel(m, d)
    g.compile(loss=["mean_squared_error", "binary_crossentropy"] + ["mean_squared_error"] * 4 + [None] * 4 + ["mean_squared_error"] * 2,
              optimizer=opt)
    gl = g.train_on_batch([a, b, c] + imgs[:5], [hr, np.ones(2, np.float32)] + feats + [hr] * 4 + [hr[::-1].copy(), a])
    out.put((losses, w_gen, dl, gl, [x.copy() for x in m.get_weights()], [x.copy() for x in d.get_weights()], opt.iterations))
    dist.barrier()
    dp.shutdown()
    dist.destroy_process_group()


def _disc_weights():
    from oracle import ref_disc as rd
    return list(rd.make_disc_weights(9, 64, 64).values())


def _gan_images():
    rng = np.random.default_rng(11)
    return [rng.random((2, 64, 64, 3), dtype=np.float32) for _ in range(9)]


@pytest.mark.parametrize("direct", [True, False], ids=["library_rccl", "torch_all_reduce"])
def test_one_rank_on_rccl_equals_the_plain_process(weights7, direct):
    """The data-parallel path executed on RCCL itself (backend nccl, one rank: all this box has): generator step,
    discriminator step and combined step with the gradient all-reduced by libsr_hip's ncclAllReduce on the ctx stream
    (sr_train_allreduce / sr_rccl_allreduce_f32; the default) or through torch.distributed must give the weights of
    the plain single-process path (a one-rank mean is the identity)."""
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    p = ctx.Process(target=_rccl_worker, args=(_free_port(), q, direct))
    p.start()
    losses, w_gen, dl, gl, w_gen2, w_disc, its = q.get(timeout=300)
    p.join(timeout=60)
    assert p.exitcode == 0
    from layers import Adam, Discriminator, GeneratorCore, GeneratorModel, TecoGANModel
    core = GeneratorCore((16, 16, 3), model="tecogan", dtype="fp32", device=0)
    core.set_tensors(list(weights7.items()))
    m = GeneratorModel(core, "Generator")
    opt = Adam(lr=1e-4)
    m.compile(loss=["mean_squared_error"] * 2, loss_weights=[1., 1.], optimizer=opt)
    a, b, c, hr = _batch(2, 16, 16)
    want_l = m.train_on_batch([a, b, c], [hr, a])
    assert np.allclose(losses, want_l, rtol=1e-6)
    for x, y in zip(w_gen, m.get_weights()):
        np.testing.assert_array_equal(x, y)
    d = Discriminator((64, 64, 3), device=0).build()
    d.set_weights(_disc_weights())
    d.compile(loss=["binary_crossentropy"] + ["mean_squared_error"] * 4, loss_weights=[1., 0., 0., 0., 0.], optimizer=opt)
    imgs = _gan_images()
    want_dl = d.train_on_batch(imgs, [np.ones(2, np.float32)] + [None] * 4)
    feats = d.predict(imgs)[1:]
    g = TecoGANModel(m, d)
    g.compile(loss=["mean_squared_error", "binary_crossentropy"] + ["mean_squared_error"] * 4 + [None] * 4 + ["mean_squared_error"] * 2,
              optimizer=opt)
    want_gl = g.train_on_batch([a, b, c] + imgs[:5], [hr, np.ones(2, np.float32)] + feats + [hr] * 4 + [hr[::-1].copy(), a])
    assert np.allclose(dl, want_dl, rtol=1e-6) and np.allclose(gl, want_gl, rtol=1e-5, atol=1e-7)
    assert its == opt.iterations == 3
    for x, y in zip(w_gen2, m.get_weights()):
        np.testing.assert_array_equal(x, y)
    for x, y in zip(w_disc, d.get_weights()):
        np.testing.assert_array_equal(x, y)
