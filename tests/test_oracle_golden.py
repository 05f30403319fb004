"""Pins the CPU oracle (oracle/) to fixtures produced by the REAL reference
(tests/golden/make_golden.py, run in the dev container).  CPU only."""
import numpy as np
import torch

from oracle import regularizers_ref as R
from oracle import vgg_ref

TINY = [16, "M", 16, "M", 32, 32, "M", 32, 32, "M"]
T = lambda a: torch.from_numpy(np.asarray(a))  # noqa: E731


def close(a, b, rtol=1e-5, atol=1e-6):
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    scale = max(np.abs(b).max(), 1e-30)
    assert a.shape == b.shape
    err = np.abs(a - b).max()
    assert err <= atol * scale + rtol * scale, (err, scale)


def tiny_params(g, prefix="p", n=18):
    return [T(g["%s%d" % (prefix, i)]) for i in range(n)]


def test_g1_tiny_fwd_bwd(golden):
    g = golden("G1_vgg_fwd_bwd")
    params = tiny_params(g, "tiny_p")
    x, y = T(g["tiny_x"]), T(g["tiny_y"])
    for kind in ("ce_mean", "ce_sum", "mse_sum_zero"):
        logits, loss, grads, _ = vgg_ref.loss_and_grads(params, TINY, x, y, kind)
        close(logits, g["tiny_%s_logits" % kind])
        close(loss, g["tiny_%s_loss" % kind])
        for i, gr in enumerate(grads):
            close(gr, g["tiny_%s_g%d" % (kind, i)])


def test_g1_small_fwd_bwd(golden):
    g = golden("G1_vgg_fwd_bwd")
    cfg = vgg_ref.CFGS["small_VGG9"]
    params = vgg_ref.init_params(cfg, (128, 128), 20, 64, np.random.RandomState(21))
    gen = np.random.RandomState(22)
    x = T(gen.standard_normal((4, 3, 64, 64)).astype(np.float32))
    y = T(gen.randint(0, 20, size=(4,)).astype(np.int64))
    logits, loss, grads, _ = vgg_ref.loss_and_grads(params, cfg, x, y, "ce_mean")
    close(logits, g["small_logits"])
    close(loss, g["small_loss"])
    assert sum(p.numel() for p in params) == 615380  # small_VGG9_cl_128_128
    for i, gr in enumerate(grads):
        gd = gr.double()
        st = np.array([gd.sum().item(), gd.abs().sum().item(), gd.pow(2).sum().sqrt().item()])
        np.testing.assert_allclose(st[1:], g["small_g%d_stats" % i][1:], rtol=1e-5)
        close(gr.flatten()[:64], g["small_g%d_head" % i], rtol=1e-4)


def test_g2_ewc_fisher(golden):
    g = golden("G2_ewc_fisher")
    params = tiny_params(g)
    prev = [torch.zeros_like(p) for p in params]
    for task in range(2):
        batches = [(T(g["t%d_x%d" % (task, b)]), T(g["t%d_y%d" % (task, b)])) for b in range(3)]
        new = R.diag_fisher(params, TINY, batches, 24)
        acc = [a + b for a, b in zip(prev, new)]   # accumelate_reg_params main_EWC.py:205-232
        for i, o in enumerate(acc):
            close(o, g["t%d_omega%d" % (task, i)], rtol=2e-5)
        prev = acc


def test_g3_mas_omega_short_last_batch(golden):
    g = golden("G3_mas_omega")
    params = tiny_params(g)
    batches = [(T(g["x%d" % b]), T(g["y%d" % b])) for b in range(3)]
    assert [b[0].shape[0] for b in batches] == [8, 8, 5]
    omega = R.mas_importance(params, TINY, batches)
    for i, o in enumerate(omega):
        close(o, g["omega%d" % i], rtol=2e-5)


def test_g4_si_steps_and_consolidation(golden):
    g = golden("G4_si")
    for tag, wd in (("wd0", 0.0), ("wd1", 1e-4)):
        theta = tiny_params(g, tag + "_p")
        omega = tiny_params(g, tag + "_omega")
        init = tiny_params(g, tag + "_init")
        w = [torch.zeros_like(t) for t in theta]
        buf = [None] * len(theta)
        for s in range(3):
            x, y = T(g["%s_x%d" % (tag, s)]), T(g["%s_y%d" % (tag, s)])
            _, _, grads, _ = vgg_ref.loss_and_grads(theta, TINY, x, y, "ce_mean")
            for i in range(len(theta)):
                theta[i], buf[i], w[i] = R.si_step(theta[i], grads[i], omega[i], init[i], w[i], buf[i],
                                                   400, 1e-2, 0.9, wd, s == 0)
        for i in range(len(theta)):
            close(theta[i], g["%s_s2_theta%d" % (tag, i)], rtol=2e-5)
            close(buf[i], g["%s_s2_buf%d" % (tag, i)], rtol=2e-4)
            close(w[i], g["%s_s2_w%d" % (tag, i)], rtol=2e-4)
            o, w0, iv = R.si_consolidate(omega[i], w[i], theta[i], init[i])
            close(o, g["%s_cons_omega%d" % (tag, i)], rtol=2e-4)
            close(iv, g["%s_cons_init%d" % (tag, i)], rtol=2e-5)
            assert float(T(g["%s_cons_w%d" % (tag, i)]).abs().max()) == 0.0


def test_g5_penalised_sgd(golden):
    g = golden("G5_reg_sgd")
    for tag, lam, wd in (("ewc", 400, 0.0), ("mas", 3, 5e-4)):
        theta = tiny_params(g, tag + "_p")
        n = len(theta)
        omega = [T(g["%s_omega%d" % (tag, i)]) if i < n - 2 else None for i in range(n)]
        init = [T(g["%s_init%d" % (tag, i)]) if i < n - 2 else None for i in range(n)]
        buf = [None] * n
        for s in range(3):
            x, y = T(g["%s_x%d" % (tag, s)]), T(g["%s_y%d" % (tag, s)])
            _, loss, grads, _ = vgg_ref.loss_and_grads(theta, TINY, x, y, "ce_mean")
            close(loss, g["%s_s%d_loss" % (tag, s)], rtol=2e-5)
            for i in range(n):
                theta[i], buf[i] = R.reg_sgd_step(theta[i], grads[i], omega[i], init[i], buf[i],
                                                  lam, 1e-2, 0.9, wd, s == 0)
        for i in range(n):
            close(theta[i], g["%s_s2_theta%d" % (tag, i)], rtol=2e-5)
            close(buf[i], g["%s_s2_buf%d" % (tag, i)], rtol=2e-4)


def test_g9_set_lr_traces(golden):
    g = golden("G9_schedules")
    for tag in ("ewc", "si"):
        for case in range(3):
            tr = R.set_lr_trace(list(g["%s_c%d_improved" % (tag, case)]), 0.01, tag)
            ref = g["%s_c%d_trace" % (tag, case)]
            assert len(tr) == len(ref)
            for (ep, lr, cont), r in zip(tr, ref):
                assert ep == int(r[0]) and cont == bool(r[2])
                assert abs(lr - r[1]) <= 1e-12 * max(1.0, r[1])


def test_g7_packnet_masks_bit_exact(golden):
    """SparsePruner / PacknetSGD (reference) vs oracle/packnet_ref.py: masks bit-exact, weights exact."""
    from oracle import packnet_ref as P
    g = golden("G7_packnet")
    layers = [int(i) for i in g["layer_idx"]]
    w = {i: g["init_w%d" % i].copy() for i in layers}
    b = {i: g["init_b%d" % i].copy() for i in layers}
    masks = {i: np.zeros(w[i].shape, dtype=np.uint8) for i in layers}
    for task, perc in ((1, 0.75), (2, 0.5)):
        masks = {i: P.make_finetuning_mask(masks[i], task) for i in layers}
        for i in layers:
            assert np.array_equal(masks[i], g["t%d_ft_m%d" % (task, i)])
        buf = {i: None for i in layers}
        bbuf = {i: None for i in layers}
        for s in range(2):
            for i in layers:
                gr = P.make_grads_zero(g["t%d_s%d_rawg%d" % (task, s, i)], masks[i], task)
                # the reference adds the weight-decay term to .grad in place (d_p.add_) before we could
                # snapshot it, so compare the exact zero set here and the weights below
                assert np.array_equal(gr == 0, g["t%d_s%d_g%d" % (task, s, i)] == 0)
                w[i], buf[i] = P.packnet_sgd_step(w[i], gr, buf[i], 0.05, 0.9, 5e-4, s == 0)
                w[i] = P.make_pruned_zero(w[i], masks[i])
                np.testing.assert_allclose(w[i], g["t%d_s%d_w%d" % (task, s, i)], rtol=1e-6, atol=1e-8)
                # biases: train_bias=False -> grads zeroed -> wd term masked out too -> unchanged
                gb = np.zeros_like(b[i])
                b[i], bbuf[i] = P.packnet_sgd_step(b[i], gb, bbuf[i], 0.05, 0.9, 5e-4, s == 0)
                assert np.array_equal(b[i], g["t%d_s%d_b%d" % (task, s, i)])
            w = {i: g["t%d_s%d_w%d" % (task, s, i)].copy() for i in layers}   # stay bit-aligned with the fixture
        for i in layers:
            w[i], masks[i], cutoff, k = P.prune(w[i], masks[i], task, perc)
            assert np.array_equal(masks[i], g["t%d_pruned_m%d" % (task, i)]), "mask layer %d task %d" % (i, task)
            assert np.array_equal(w[i], g["t%d_pruned_w%d" % (task, i)])
    for i in layers:
        assert np.array_equal(P.apply_mask(w[i], masks[i], 1), g["apply1_w%d" % i])
    # banker's rounding of the cutoff rank (prune.py:32)
    assert P.cutoff_rank(0.5, 5) == 2 and P.cutoff_rank(0.5, 7) == 4 and P.cutoff_rank(0.9, 215) == 194


def _hat_params(g, prefix="p_"):
    return {str(n): T(g[prefix + str(n)]) for n in g["param_names"]}


def test_g8_hat_oracle(golden):
    from oracle import hat_ref as H
    g = golden("G8_hat")
    smax, lamb, t, lr, mom, wd = [float(v) for v in g["hyper"]]
    t = int(t)
    P = _hat_params(g)
    pool_after = {0, 1, 3, 5}        # TINY = [16,'M',16,'M',32,32,'M',32,32,'M']
    mask_pre, mask_back = H.init_masks(P, t, smax)
    for i, mp in enumerate(mask_pre):
        close(mp, g["mask_pre%d" % i].reshape(-1), rtol=1e-6)
    for n, v in mask_back.items():
        close(v, g["mask_back_" + n], rtol=1e-6)
    assert set(mask_back) == {k[len("mask_back_"):] for k in g.files if k.startswith("mask_back_")}
    bufs = {n: None for n in P}
    for step, s in enumerate((7.3, 23.0)):
        x, y = T(g["x%d" % step]), T(g["y%d" % step])
        leaf = {n: v.clone().requires_grad_(True) for n, v in P.items()}
        logits, mk = H.forward(leaf, pool_after, t, x, s)
        loss, reg = H.criterion(logits, y, mk, mask_pre, lamb)
        loss.backward()
        close(logits.detach(), g["s%d_logits" % step], rtol=2e-5)
        close(loss.detach(), g["s%d_loss" % step], rtol=2e-5)
        close(reg.detach(), g["s%d_reg" % step], rtol=2e-5)
        for n in P:
            if leaf[n].grad is None:
                continue
            P[n], bufs[n], _ = H.hat_sgd_step(n, P[n], leaf[n].grad, bufs[n], mask_back, t, s, smax, lr, mom, wd,
                                              first=(step == 0))
            if "embs" in n:
                P[n] = torch.clamp(P[n], -6, 6)
            close(P[n], g["s%d_theta_%s" % (step, n)], rtol=5e-5)


def test_g6_gem_oracle_and_product_qp(golden):
    """gem.py store/overwrite/project2cone2 (reference, QP via scipy stand-in) vs the oracle (active-set
    enumeration) and the product's Goldfarb-Idnani solver.  quadprog parity itself is unpinned."""
    from oracle import gem_ref as GR
    from oracle import qp_ref as qp
    g = golden("G6_gem")
    shapes = [(4, 3, 3, 3), (4,), (6, 4), (6,)]
    G = np.zeros_like(g["G"])
    for tid in (0, 2, 3):
        G = GR.store_grad([g["grad_t%d_%d" % (tid, i)] for i in range(4)], G, tid)
    assert np.array_equal(G, g["G"])
    for i, a in enumerate(GR.overwrite_grad(G[:, 2] * 2.0, shapes)):
        assert np.array_equal(a, g["overwritten_%d" % i])
    for case in range(6):
        mem, gr, margin = g["qp%d_mem" % case], g["qp%d_g" % case], float(g["qp%d_margin" % case])
        x, v = GR.project2cone2(gr, mem, margin)
        scale = np.abs(g["qp%d_x" % case]).max()
        assert np.abs(x - g["qp%d_x" % case].reshape(-1)).max() <= 2e-5 * scale     # SLSQP stand-in tolerance
        # product solver from the Gram matrix (what the device computes)
        rows = np.concatenate([mem.T, gr.reshape(1, -1)]).astype(np.float64)
        gram = rows @ rows.T
        t = mem.shape[1]
        v2 = qp.project2cone2_coefficients(gram, t, list(range(t)), margin)
        assert np.abs(v2 - v).max() <= 1e-9 * max(1.0, np.abs(v).max())
        # KKT of the product solution
        M = mem.T.astype(np.float64)
        Pm = M @ M.T
        Pm = 0.5 * (Pm + Pm.T) + 1e-3 * np.eye(t)
        grad = Pm @ v2 + M @ gr.reshape(-1).astype(np.float64)
        assert np.all(v2 >= margin - 1e-10) and np.all(grad >= -1e-8)
        assert abs(((v2 - margin) * grad).sum()) <= 1e-7 * max(1.0, np.abs(grad).max())


def _g13_models(g):
    out = []
    for seed in g["model_seeds"]:
        ps = vgg_ref.init_params(TINY, (24, 24), 5, 32, np.random.RandomState(int(seed)))
        for i in (-6, -4, -2):
            ps[i] = ps[i] * 20.0
        out.append(ps)
    return out


def test_g13_imm_merge_and_precision(golden):
    """oracle/imm_ref.py vs the reference's IMM_merge_models (mean / mode) and diag_fisher (G13, make_g13.py)."""
    from oracle import imm_ref as I
    g = golden("G13_imm")
    models = _g13_models(g)
    keep = (0, 1, 6, 7, 12, 13)
    for idx in (1, 2):
        for j in keep:
            thetas = [m[j] for m in models[:idx + 1]]
            assert torch.equal(I.merge_mean(thetas), torch.from_numpy(g["mean%d_p%d" % (idx, j)])), ("mean", idx, j)
            precs = [torch.from_numpy(g["prec%d_p%d" % (i, j)]) for i in range(idx + 1)]
            s = precs[0]
            for p in precs[1:]:
                s = s + p
            np.testing.assert_allclose(I.merge_mode(thetas, precs, s).numpy(), g["mode%d_p%d" % (idx, j)], rtol=1e-6, atol=1e-7)
        for j in (16, 17):      # heads are not merged: the task's own head
            assert torch.equal(models[idx][j], torch.from_numpy(g["mean%d_p%d" % (idx, j)]))
            assert torch.equal(models[idx][j], torch.from_numpy(g["mode%d_p%d" % (idx, j)]))
    phases = [[torch.from_numpy(g["fx_train%d" % b]) for b in range(2)], [torch.from_numpy(g["fx_val%d" % b]) for b in range(3)]]
    with torch.no_grad():
        targets = [[vgg_ref.forward(models[0], TINY, x).argmax(1) for x in ph] for ph in phases]
    prec = I.diag_fisher(models[0], TINY, phases, targets, exclude=(16, 17))
    for j in range(16):
        ref = g["fisher_p%d" % j]
        np.testing.assert_allclose(prec[j].numpy(), ref, rtol=2e-4, atol=2e-4 * float(np.abs(ref).max()))
    assert "fisher_p16" not in g.files and prec[16] is None
    assert int(g["urp_count"]) == 18 and bool(g["urp_omega_all_ones"]) and bool(g["urp_init_equals_theta"])


def test_g14_lwf_distillation(golden):
    """oracle/lwf_ref.py vs the reference's distillation_loss value + gradient (G14, make_g14.py)."""
    from oracle import lwf_ref as LW
    g = golden("G14_lwf")
    for tag in ("a", "b"):
        y = torch.from_numpy(g["d%s_y" % tag]).requires_grad_(True)
        loss = LW.distillation_loss(y, torch.from_numpy(g["d%s_t" % tag]), float(g["d%s_T" % tag]))
        loss.backward()
        np.testing.assert_allclose(loss.detach().numpy(), g["d%s_loss" % tag], rtol=1e-5)
        np.testing.assert_allclose(y.grad.numpy(), g["d%s_grad" % tag], rtol=1e-4, atol=1e-7)
    outs = [torch.from_numpy(g["out%d" % i]) for i in range(3)]
    task, dist = LW.lwf_objective(outs, torch.from_numpy(g["y"]), [torch.from_numpy(g["teacher0"]), torch.from_numpy(g["teacher1"])], 2.0, 10.0)
    np.testing.assert_allclose(task.numpy(), g["task_loss"], rtol=1e-5)
    np.testing.assert_allclose(dist.numpy(), g["dist_loss"], rtol=1e-5)


def _g15_net(g, prefix="p0_"):
    import os
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden"))
    import g15_inputs as I
    m = I.SmallAlexNet(num_classes=I.N_OUT)
    I.load_params(m, [g["%s%d" % (prefix, i)] for i in range(len(list(m.parameters())))])
    return m, I


def test_g15_gem_alexnet_dropout_observe(golden):
    """oracle/alexnet_ref.py (manual dropout rows, memory pass + current batch under ONE mask set, violation test,
    projection, momentum SGD) vs the reference's unchanged gem.Net on an AlexNet-structured net (G15, make_g15.py)."""
    from oracle import alexnet_ref as AR
    g = golden("G15_gem_alexnet")
    m, I = _g15_net(g)
    data = [(torch.from_numpy(x), torch.from_numpy(y)) for x, y in I.batches(steps=6)]
    bufs = [None] * len(list(m.parameters()))
    cum = [4, 8]
    assert int(g["s2_proj"]) == 1                      # the fixture holds a projected step
    for step in range(4):
        t = 0 if step < 2 else 1
        masks = {0: torch.from_numpy(g["s%d_mask0" % step])[None], 1: torch.from_numpy(g["s%d_mask3" % step])[None]}
        # ring buffer of task 0 after two observes: the first n_memories samples of step 1 (gem.py:322-345)
        mem = [(0, [(data[1][0][:I.N_MEM], data[1][1][:I.N_MEM])])] if t == 1 else []
        x, y = data[step]
        loss, hits, viol, G = AR.gem_observe(m, bufs, x, t, y, masks, mem, cum, 0.002, 0.5)
        np.testing.assert_allclose(float(loss), float(g["s%d_loss" % step]), rtol=2e-5)
        assert hits == int(g["s%d_hits" % step]) and viol == int(g["s%d_proj" % step])
    np.testing.assert_array_equal(g["s1_mem_labels"][0], data[1][1][:I.N_MEM].numpy())
    for i, p in enumerate(m.parameters()):
        np.testing.assert_allclose(p.detach().numpy(), g["p4_%d" % i], rtol=2e-4, atol=2e-6)
    # observe_FT after init_setup: fresh momentum, ONE mask set for both steps
    np.testing.assert_array_equal(g["s4_mask3"], g["s5_mask3"])
    assert not np.array_equal(g["s4_mask3"], g["s3_mask3"])
    bufs = [None] * len(bufs)
    masks = {0: torch.from_numpy(g["s4_mask0"])[None], 1: torch.from_numpy(g["s4_mask3"])[None]}
    params = list(m.parameters())
    for step in (4, 5):
        x, y = data[step]
        loss = torch.nn.functional.cross_entropy(AR.forward(m, x, masks)[:, 4:8], y)
        np.testing.assert_allclose(float(loss), float(g["s%d_loss" % step]), rtol=2e-5)
        AR.sgd_momentum_step(params, [gr.detach() for gr in torch.autograd.grad(loss, params)], bufs, 0.002)
    for i, p in enumerate(params):
        np.testing.assert_allclose(p.detach().numpy(), g["p6_%d" % i], rtol=2e-4, atol=2e-6)
    with torch.no_grad():
        np.testing.assert_allclose(AR.forward(m, data[0][0]).numpy()[:, 4:8], g["eval_logits_t1"][:, 4:8], rtol=1e-4, atol=1e-5)
    assert (g["eval_logits_t1"][:, :4] < -1e10).all()


TINY16 = [16, "M", 16, "M", 32, 32, "M", 32, 32, "M"]


def _g16_params(seed):
    params = vgg_ref.init_params(TINY16, (24, 24), 4, 32, np.random.RandomState(seed))
    for i in (12, 14, 16):               # make_g16.tiny_net scales the three Linear weights by 20
        params[i] = params[i] * 20.0
    return params


def test_g16_ebll_objectives(golden):
    """oracle/ebll_ref.py vs the reference's EBLL code (G16, make_g16.py): autoencoder stage (losses, the four
    gradients, three Adadelta steps) and the EBLL objective with every feature-extractor / classifier gradient."""
    from oracle import ebll_ref as EB
    g = golden("G16_ebll")
    x, y = torch.from_numpy(g["x"]), torch.from_numpy(g["y"])
    # ---- stage 1
    params = _g16_params(161)
    feat = vgg_ref.features(params, TINY16, x).flatten(1)
    np.testing.assert_allclose(feat.numpy(), g["s1_in"], rtol=1e-5, atol=1e-6)
    ae = [torch.from_numpy(g["s1_" + n]).clone().requires_grad_(True)
          for n in ("encode.0.weight", "encode.0.bias", "decode.0.weight", "decode.0.bias")]

    def tail(r):
        h = torch.relu(torch.nn.functional.linear(r, params[12], params[13]))
        h = torch.relu(torch.nn.functional.linear(h, params[14], params[15]))
        return torch.nn.functional.linear(h, params[16], params[17])

    state = [(torch.zeros_like(p), torch.zeros_like(p)) for p in ae]
    for step in range(3):
        task, enc, recon = EB.stage1_objective(feat, y, ae, tail, float(g["s1_alpha"]))
        grads = torch.autograd.grad(float(g["s1_alpha"]) * enc + task, ae)
        if step == 0:
            np.testing.assert_allclose(task.item(), g["s1_task_loss"], rtol=1e-5)
            np.testing.assert_allclose(enc.item(), g["s1_enc_loss"], rtol=1e-5)
            np.testing.assert_allclose(recon.detach().numpy(), g["s1_recon"], rtol=1e-5, atol=1e-6)
            for p_grad, n in zip(grads, ("encode.0.weight", "encode.0.bias", "decode.0.weight", "decode.0.bias")):
                np.testing.assert_allclose(p_grad.numpy(), g["s1_grad_" + n], rtol=1e-4, atol=1e-7)
        EB.adadelta_step(ae, grads, state, 0.01)
    for p, n in zip(ae, ("encode.0.weight", "encode.0.bias", "decode.0.weight", "decode.0.bias")):
        np.testing.assert_allclose(p.detach().numpy(), g["s1_after3_" + n], rtol=1e-5, atol=1e-7)
    # ---- stage 2
    params = [p.clone().requires_grad_(True) for p in _g16_params(162)]
    heads = [(torch.from_numpy(g["s2_head%d_w" % i]).requires_grad_(True), torch.from_numpy(g["s2_head%d_b" % i]).requires_grad_(True))
             for i in (1, 2)]
    feat = vgg_ref.features(params, TINY16, x).flatten(1)
    h = torch.relu(torch.nn.functional.linear(feat, params[12], params[13]))
    h = torch.relu(torch.nn.functional.linear(h, params[14], params[15]))
    outs = [torch.nn.functional.linear(h, params[16], params[17])] + [torch.nn.functional.linear(h, w, b) for w, b in heads]
    codes = [EB.encode(feat, torch.from_numpy(g["s2_ae%d_encode.0.weight" % i]), torch.from_numpy(g["s2_ae%d_encode.0.bias" % i]))
             for i in (0, 1)]
    for i in range(3):
        np.testing.assert_allclose(outs[i].detach().numpy(), g["s2_out%d" % i], rtol=1e-4, atol=1e-5)
    for i in range(2):
        np.testing.assert_allclose(codes[i].detach().numpy(), g["s2_code%d" % i], rtol=1e-5, atol=1e-6)
    task, dist, code = EB.stage2_objective(outs, codes, y, [torch.from_numpy(g["s2_tlogits%d" % i]) for i in (0, 1)],
                                           [torch.from_numpy(g["s2_tcodes%d" % i]) for i in (0, 1)], 2.0,
                                           float(g["s2_lambda"]), float(g["s2_reg_alpha"]))
    np.testing.assert_allclose(task.item(), g["s2_task_loss"], rtol=1e-5)
    np.testing.assert_allclose(dist.item(), g["s2_dist_loss"], rtol=1e-5)
    np.testing.assert_allclose(code.item(), g["s2_code_loss"], rtol=1e-5)
    leaves = params + [t for wb in heads for t in wb]
    grads = torch.autograd.grad(task + dist + float(g["s2_reg_alpha"]) * code, leaves)
    assert len(grads) == len(g["s2_param_names"])
    for j, gr in enumerate(grads):
        ref = g["s2_g%d" % j]
        assert np.abs(gr.numpy() - ref).max() <= 1e-4 * max(np.abs(ref).max(), 1e-6), g["s2_param_names"][j]


def test_g19_oracle_importance_on_the_reference_checkpoint(golden):
    """The oracle's diag_fisher / mas_importance on the first-task checkpoint the reference's framework trained, against the
    Omega tensors the reference's own accumulate_EWC_weights / accumulate_objective_based_weights produced from it (G19)."""
    import os
    import tempfile
    from clsurvey_amd.framework.tasks import SyntheticTaskSequence
    g = golden("G19_teacher_forced_omega")
    cfg = vgg_ref.CFGS["small_VGG9"]
    theta = [T(g["theta%d" % i]) for i in range(18)]
    with tempfile.TemporaryDirectory() as root:
        ds = SyntheticTaskSequence(os.path.join(root, "data"), task_count=3, classes_per_task=4, sizes=(160, 40, 40), hw=32,
                                   noise=0.4, name="tiny3")
        d = torch.load(ds.get_task_dataset_path("1"), weights_only=False)["train"]
    batches = [(d.x[i:i + 40], d.y[i:i + 40]) for i in range(0, 160, 40)]
    for tag, omega in (("ewc", R.diag_fisher(theta, cfg, batches, 160)), ("mas", R.mas_importance(theta, cfg, batches))):
        for i, o in enumerate(omega):
            flat = o.numpy().reshape(-1)
            if flat.size > (1 << 16):
                flat = flat[np.sort(np.random.RandomState(19).choice(flat.size, 8192, replace=False))]
            ref = g["%s_omega%d" % (tag, i)]
            assert float(np.abs(flat - ref).max()) <= 2e-5 * float(g["%s_stats%d" % (tag, i)][1]), (tag, i)


def test_g20_hat_oracle_at_wide_vgg9_widths(golden):
    """The HAT oracle at config 5's real widths (wide_VGG9_cl_512_512, 3x64x64, batch 8) against the reference's
    unchanged vgg_hat.Net / Appr.criterion / HAT_SGD run of fixture G20: logits, loss, gates in full, gradients and
    updated parameters at the fixture's sampled positions with their float64 checksums.  First step only (s = 3.1):
    the second enters with parameters that already differ in the last bits (see tests/test_gpu_wide.py)."""
    import os
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden"))
    import g20_common as C
    from oracle import hat_ref as H
    g = golden("G20_wide_widths")
    smax, lamb, t, lr, mom, wd = [float(v) for v in g["hat64_hyper"]]
    t = int(t)
    names = [str(n) for n in g["hat64_param_names"]]
    widths = [v for v in C.WIDE if v != "M"]
    pool_after, n_conv = set(), 0
    for v in C.WIDE:
        if v == "M":
            pool_after.add(n_conv - 1)
        else:
            n_conv += 1
    assert pool_after == {0, 1, 3, 5}
    fc_in = widths[-1] * (64 // 16) ** 2

    def shape(n):
        kind, idx, leaf = n.split(".")
        i = int(idx)
        if kind == "convs":
            return (widths[i], 3 if i == 0 else widths[i - 1], 3, 3) if leaf == "weight" else (widths[i],)
        if kind == "conv_embs":
            return (3, widths[i])
        if kind == "fcs":
            return (C.FC[i], fc_in if i == 0 else C.FC[i - 1]) if leaf == "weight" else (C.FC[i],)
        if kind == "fc_embs":
            return (3, C.FC[i])
        return (C.NCLS, C.FC[-1]) if leaf == "weight" else (C.NCLS,)

    P = {n: T(a) for n, a in zip(names, C.fill_params([(n, shape(n)) for n in names], 2001))}
    mask_pre, mask_back = H.init_masks(P, t, smax)
    x, y = (T(a) for a in C.batch(2100, 8, 64))
    leaf = {n: v.clone().requires_grad_(True) for n, v in P.items()}
    logits, mk = H.forward(leaf, pool_after, t, x, 3.1)
    loss, reg = H.criterion(logits, y, mk, mask_pre, lamb)
    loss.backward()
    close(logits.detach(), g["hat64_s0_logits"], rtol=1e-4)
    loss_ref, reg_ref = g["hat64_s0_loss"]
    assert abs(float(loss.detach()) - loss_ref) <= 1e-5 * abs(loss_ref) and abs(float(reg.detach()) - reg_ref) <= 1e-5 * abs(reg_ref) + 1e-7
    for i, m in enumerate(mk):
        close(m.detach(), g["hat64_s0_mask%d" % i].reshape(-1), rtol=1e-6)

    def check(tag, tensor, seed, tol):
        d = C.digest(tensor.detach().numpy(), seed)
        ref_v, ref_s = g[tag + "__v"], g[tag + "__s"]
        assert d["v"].shape == ref_v.shape and d["s"][2] == ref_s[2], tag
        assert np.abs(d["v"] - ref_v).max() <= tol * max(np.abs(ref_v).max(), 1e-30), tag
        assert abs(d["s"][0] - ref_s[0]) <= tol * max(ref_s[1], 1e-30), tag

    n_grad = 0
    for j, n in enumerate(names):
        if "hat64_s0_grad_%s__v" % n in g.files:
            check("hat64_s0_grad_" + n, leaf[n].grad, 2200 + j, 2e-4)
            n_grad += 1
        new, _, _ = H.hat_sgd_step(n, P[n], leaf[n].grad, None, mask_back, t, 3.1, smax, lr, mom, wd, first=True)
        if "embs" in n:
            new = torch.clamp(new, -6, 6)
        check("hat64_s0_theta_" + n, new, 2300 + j, 1e-5)
    assert n_grad == len(names)


def test_g20_packnet_oracle_at_wide_vgg9_widths(golden):
    """PackNet's batch step (methods/packnet/main.py:164-198: forward on the current head, backward, foreign gradients and
    shared biases to zero, PacknetSGD with momentum, pruned weights to zero) restated with the oracle's pieces (vgg_ref +
    packnet_ref) at wide_VGG9 widths, 3x64x64, two batches of 8, against the reference's unchanged run of fixture G20:
    updated parameters at the sampled positions / checksums, the zero bitmap of every shared layer bit-exact."""
    import os
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden"))
    import g20_common as C
    from oracle import packnet_ref as PK
    g = golden("G20_wide_widths")
    tag, seed, hw, nb = "pack64", 3000, 64, 8
    lr, mom, wd = [float(v) for v in g[tag + "_hyper"][:3]]
    names = [str(n) for n in g[tag + "_param_names"]]
    widths = [v for v in C.WIDE if v != "M"]
    seq_idx = [0, 3, 6, 8, 11, 13, 17, 19]                 # Conv2d / Linear positions inside `shared`
    assert [i + 1 for i in seq_idx] == g[tag + "_layout"].tolist()        # .modules() counts the Sequential itself

    def shape(n):
        kind, idx, leaf = n.split(".")
        if kind == "classifiers":
            return (C.NCLS, C.FC[-1]) if leaf == "weight" else (C.NCLS,)
        j = seq_idx.index(int(idx))
        if j < 6:
            return (widths[j], 3 if j == 0 else widths[j - 1], 3, 3) if leaf == "weight" else (widths[j],)
        fin = widths[-1] * (hw // 16) ** 2 if j == 6 else C.FC[0]
        return (C.FC[j - 6], fin) if leaf == "weight" else (C.FC[j - 6],)

    theta = dict(zip(names, C.fill_params([(n, shape(n)) for n in names], seed)))
    owner = {"shared.%d.weight" % i: C.owner_mask(seed + 100 + i + 1, shape("shared.%d.weight" % i)) for i in seq_idx}
    for n, m in owner.items():
        theta[n] = PK.make_pruned_zero(theta[n], m)
    frozen = {n: theta[n].copy() for n in owner}
    order = ["shared.%d.%s" % (i, leaf) for i in seq_idx for leaf in ("weight", "bias")] + ["classifiers.1.weight", "classifiers.1.bias"]
    bufs = {n: None for n in names}
    for step in range(2):
        x, y = (T(a) for a in C.batch(seed + 10 + step, nb, hw))
        logits, _, grads, _ = vgg_ref.loss_and_grads([T(theta[n]) for n in order], C.WIDE, x, y, "ce_mean")
        err = 100.0 * float((logits.argmax(1) != y).sum()) / nb
        assert abs(err - float(g["%s_s%d_err" % (tag, step)][0])) <= 1e-6
        for n, gr in zip(order, grads):
            gr = gr.numpy()
            if n in owner:
                gr = PK.make_grads_zero(gr, owner[n], 2)
            elif n.startswith("shared"):
                gr = np.zeros_like(gr)                                  # prune.py:91-93: shared biases stay fixed
            theta[n], bufs[n] = PK.packnet_sgd_step(theta[n], gr, bufs[n], lr, mom, wd, first=(step == 0))
            if n in owner:
                theta[n] = PK.make_pruned_zero(theta[n], owner[n])
        for j, n in enumerate(names):
            d = C.digest(theta[n], seed + 300 + j)
            ref_v, ref_s = g["%s_s%d_theta_%s__v" % (tag, step, n)], g["%s_s%d_theta_%s__s" % (tag, step, n)]
            assert d["v"].shape == ref_v.shape and d["s"][2] == ref_s[2], n
            assert np.abs(d["v"] - ref_v).max() <= 2e-5 * max(np.abs(ref_v).max(), 1e-30), (step, n)
            assert abs(d["s"][0] - ref_s[0]) <= 2e-5 * max(ref_s[1], 1e-30), (step, n)
            if n in owner:
                assert C.zero_pattern(theta[n]) == str(g["%s_s%d_zeros_%s" % (tag, step, n)]), n
    for n, m in owner.items():
        assert np.array_equal(theta[n][m == 1], frozen[n][m == 1]), n           # the earlier task's weights: bit-exact


def test_g21_hat_alexnet_oracle(golden):
    """The HAT oracle on AlexNet (alexnet_hat.Net: 11x11/4, 5x5, 3x3 convolutions, 3x3/2 max-pools, Dropout in FRONT of
    each gated Linear layer) at 3x224x224 against the reference's unchanged code (fixture G21): eval-mode logits and gates
    at s = smax; one training step under the fixture's Dropout masks — logits, loss, sampled gradients and updated
    parameters.  Same ATen kernels on both sides: 1e-5, no decision flips."""
    import os
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden"))
    import g20_common as C
    from oracle import hat_ref as H
    g = golden("G21_hat_alexnet")
    smax, lamb, t, lr, mom, wd = [float(v) for v in g["hyper"]]
    t, nb = int(t), 4
    names = [str(n) for n in g["param_names"]]
    convs = [(64, 3, 11), (192, 64, 5), (384, 192, 3), (256, 384, 3), (256, 256, 3)]       # models/net.py:96-125 (torchvision AlexNet)
    geo = [(4, 2), (1, 2), (1, 1), (1, 1), (1, 1)]
    fcs = [(4096, 256 * 6 * 6), (4096, 4096)]

    def shape(n):
        kind, idx, leaf = n.split(".")
        i = int(idx)
        if kind == "convs":
            return (convs[i][0], convs[i][1], convs[i][2], convs[i][2]) if leaf == "weight" else (convs[i][0],)
        if kind == "conv_embs":
            return (3, convs[i][0])
        if kind == "fcs":
            return fcs[i] if leaf == "weight" else (fcs[i][0],)
        if kind == "fc_embs":
            return (3, fcs[i][0])
        return (C.NCLS, 4096) if leaf == "weight" else (C.NCLS,)

    P = {n: T(a) for n, a in zip(names, C.fill_params([(n, shape(n)) for n in names], 5001))}
    kw = dict(conv_geo=geo, pool=(3, 2), first_drop=True)
    x, y = (T(a) for a in C.batch(5100, nb, 224))
    with torch.no_grad():
        logits, mk = H.forward(P, {0, 1, 4}, t, x, smax, **kw)
    close(logits, g["eval_logits"], rtol=1e-5)
    for i, m in enumerate(mk):
        close(m, g["eval_mask%d" % i].reshape(-1), rtol=1e-6)
    # one training step under the reference's Dropout masks
    mask_pre, mask_back = H.init_masks(P, t, smax)
    gen = np.random.RandomState(5200)
    drop = [T((gen.rand(nb, d) < 0.5).astype(np.float32) * 2.0) for d in (256 * 6 * 6, 4096)]
    loss_ref, reg_ref, s = [float(v) for v in g["train_loss"]]
    x, y = (T(a) for a in C.batch(5101, nb, 224))
    leaf = {n: v.clone().requires_grad_(True) for n, v in P.items()}
    logits, mk = H.forward(leaf, {0, 1, 4}, t, x, s, drop=drop, **kw)
    loss, reg = H.criterion(logits, y, mk, mask_pre, lamb)
    loss.backward()
    close(logits.detach(), g["train_logits"], rtol=1e-5)
    assert abs(float(loss.detach()) - loss_ref) <= 1e-5 * abs(loss_ref) and abs(float(reg.detach()) - reg_ref) <= 1e-5 * abs(reg_ref) + 1e-7

    def check(tag, tensor, seed, tol):
        d = C.digest(tensor.detach().numpy(), seed)
        ref_v, ref_s = g[tag + "__v"], g[tag + "__s"]
        assert d["v"].shape == ref_v.shape and d["s"][2] == ref_s[2], tag
        assert np.abs(d["v"] - ref_v).max() <= tol * max(np.abs(ref_v).max(), 1e-30), tag
        assert abs(d["s"][0] - ref_s[0]) <= tol * max(ref_s[1], 1e-30), tag

    for j, n in enumerate(names):
        if "train_grad_%s__v" % n in g.files:
            check("train_grad_" + n, leaf[n].grad, 5300 + j, 1e-4)
        new, _, _ = H.hat_sgd_step(n, P[n], leaf[n].grad, None, mask_back, t, s, smax, lr, mom, wd, first=True)
        if "embs" in n:
            new = torch.clamp(new, -6, 6)
        check("train_theta_" + n, new, 5400 + j, 1e-5)


def test_model_name_training_step_oracle_matches_reference_g34(golden):
    """G34 (tests/golden/make_g34.py): the reference's factory (models/net.py:15-36, VGGSlim.py:27-76) built the `_BN`, `_DROP`,
    `_DROP_BN` and deep_VGG22 models and ran one TRAINING-MODE step on them.  The torch-CPU oracle (oracle/alexnet_ref.py: a
    walk over features / classifier with Dropout masks as data) on the BUILD's `parse_model_name` model of the same name, same
    seeded parameters, batch and masks, reproduces the reference's logits, loss, every gradient and the BatchNorm buffers —
    so module order, BatchNorm / Dropout placement and train-mode semantics of the build's models are the reference's."""
    import copy
    import sys
    import numpy as np
    import torch
    import os
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden"))
    import g20_common as GC
    from clsurvey_amd import models
    from oracle import alexnet_ref
    g = golden("G34_model_names")
    names = ["small_VGG9_cl_128_128_BN", "small_VGG9_cl_128_128_DROP", "small_VGG9_cl_128_128_DROP_BN", "deep_VGG22_cl_512_512"]
    for k, name in enumerate(names):
        m = models.parse_model_name(name, (32, 32), 20)
        named = [(n, tuple(p.shape)) for n, p in m.named_parameters()]
        mods = dict(m.named_modules())
        with torch.no_grad():
            for (n, p), q in zip(m.named_parameters(), GC.fill_params(named, 3400 + k)):
                if isinstance(mods[n.rsplit(".", 1)[0]], torch.nn.BatchNorm2d):
                    q = (1.0 + 2.0 * q) if n.endswith("weight") else q
                p.copy_(torch.from_numpy(q))
        x, y = (torch.from_numpy(a) for a in GC.batch(3450 + k, 6, 32, 20))
        masks = {i: torch.from_numpy(g[name + "__dropmask%d" % i]) for i in range(sum(1 for f in g.files if f.startswith(name + "__dropmask")))}
        m.train()
        loss, logits, grads = alexnet_ref.loss_and_grads(m, x, y, masks or None)
        ref_logits = g[name + "__logits"]
        assert float(np.abs(logits.numpy() - ref_logits).max()) <= 1e-5 * float(np.abs(ref_logits).max()), name
        assert abs(float(loss) - float(g[name + "__loss"][0])) <= 1e-5 * abs(float(g[name + "__loss"][0]))
        gmax = max(float(t.abs().max()) for t in grads)
        for j, ((n, _), t) in enumerate(zip(named, grads)):
            v, sums = g["%s__grad_%s__v" % (name, n)], g["%s__grad_%s__s" % (name, n)]
            got = t.numpy().reshape(-1)
            assert got.size == int(sums[2])
            e = float(np.abs(got[GC.positions(got.size, 3500 + 50 * k + j)] - v).max()) / max(float(np.abs(v).max()), 1e-4 * gmax)
            assert e <= 1e-5, (name, n, e)
            assert abs(float(got.astype(np.float64).sum()) - float(sums[0])) <= 1e-5 * float(sums[1]) + 1e-12, (name, n)
        for n, b in m.named_buffers():
            ref = g["%s__buf_%s" % (name, n)]
            if b.dtype == torch.float32:
                assert float((b - torch.from_numpy(ref)).abs().max()) <= 1e-6 * max(1.0, float(np.abs(ref).max())), (name, n)
