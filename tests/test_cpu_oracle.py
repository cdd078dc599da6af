"""CPU tests (-m "not gpu"): the oracle against the committed golden vectors (generated from an import of the
reference by oracle/gen_golden.py) and against the reference test-suite's known answers."""
import os

import numpy as np
import pytest
import torch

from oracle import prithvi_oracle as O
from oracle.cases import CASES, GRAD_KEYS, case_config, class_weights_for, make_inputs, sub

GOLD = os.path.join(os.path.dirname(__file__), "golden")


def test_pos_embed_tables_match_reference_fixture():
    g = np.load(os.path.join(GOLD, "pos_embed.npz"))
    for D in (256, 768, 1024):
        for T in (1, 3):
            pe = O.sincos_pos_embed_3d(D, (T, 14, 14), True).astype(np.float32)
            assert pe.shape == (1 + T * 196, D)
            assert np.all(pe[0] == 0)  # cls row is zeros (tests/model_tests/test_model.py:53)
            assert abs(pe.astype(np.float64).sum() - float(g[f"D{D}_T{T}_sum"])) < 1e-6
            assert np.array_equal(pe[[0, 1, 2, 15, 196, pe.shape[0] - 1]], g[f"D{D}_T{T}_rows"])


@pytest.mark.parametrize("name", ["tiny_t1_c2", "tiny_t3_c13"])
def test_oracle_eval_forward_matches_golden(name):
    variant, T, ncls, B, depth = CASES[name]
    cfg = case_config(name)
    sd = O.make_state_dict(cfg, seed=1042)
    img, _ = make_inputs(name, cfg, B)
    g = np.load(os.path.join(GOLD, f"{name}.npz"))
    stages = {}
    with torch.no_grad():
        logits = O.prithvi_seg_forward(cfg, sd, img, training=False, stages=stages)
    assert logits.shape == (B, ncls, 224, 224)
    assert np.abs(sub(logits) - g["eval_logits_sub"]).max() < 5e-5
    assert np.abs(sub(stages["features"]) - g["features_sub"]).max() < 5e-5
    for k in ("patch_embed", "block0", "encoder_out", "head0", "head3"):
        assert np.abs(sub(stages[k]) - g[f"stage_{k}_sub"]).max() < 1e-4, k
    assert np.array_equal(np.bincount(logits.argmax(1).reshape(-1).numpy(), minlength=ncls), g["eval_argmax_hist"])


def test_oracle_train_step_matches_golden_fp64():
    name = "tiny_t1_c2"
    cfg = case_config(name)
    sd = O.make_state_dict(cfg, seed=1042)
    img, lab = make_inputs(name, cfg, 2)
    g = np.load(os.path.join(GOLD, f"{name}.npz"))
    sd64 = {k: (v.double() if v.is_floating_point() else v) for k, v in sd.items()}
    keys = GRAD_KEYS[:3] + GRAD_KEYS[-3:]
    logits, loss, grads = O.train_step_reference(cfg, sd64, img.double(), lab, class_weights_for(2).double(), -1, keys)
    assert abs(loss.item() - float(g["train_loss"])) < 1e-10
    assert np.abs(sub(logits) - g["train_logits_sub"]).max() < 1e-9
    for k in keys:
        assert np.abs(sub(grads[k], 1024) - g["grad_sub__" + k]).max() <= 1e-9 * max(1.0, float(g["grad_norm__" + k]))
    pred = logits.argmax(1)
    cm = O.confusion_matrix(lab.numpy(), pred.numpy(), 2, -1)
    assert np.array_equal(cm, g["confusion"])
    assert abs(O.confusion_metrics(cm)["jaccard"] - float(g["miou"])) < 1e-12


def test_metrics_known_answers():
    """Reference tests/model_tests/test_metrics.py:27-74: streaming matrix == sklearn-style macro scores."""
    g = np.load(os.path.join(GOLD, "metrics.npz"))
    cm = np.zeros((3, 3), dtype=np.int64)
    for a, b in zip(np.array_split(g["y_true"], 10), np.array_split(g["y_pred"], 10)):
        cm += O.confusion_matrix(a, b, 3, None)
    assert np.array_equal(cm, g["matrix"])
    m = O.confusion_metrics(cm)
    got = np.array([m[k] for k in ("accuracy", "precision", "recall", "f1", "jaccard")])
    assert np.allclose(got, g["scalars"], atol=1e-12)
    assert np.allclose(m["jaccard_per_class"], g["jaccard_per_class"], atol=1e-12)
    from sklearn.metrics import accuracy_score, f1_score, jaccard_score, precision_score, recall_score

    yt, yp = g["y_true"], g["y_pred"]
    assert abs(m["accuracy"] - accuracy_score(yt, yp)) < 1e-12
    assert abs(m["precision"] - precision_score(yt, yp, average="macro")) < 1e-12
    assert abs(m["recall"] - recall_score(yt, yp, average="macro")) < 1e-12
    assert abs(m["f1"] - f1_score(yt, yp, average="macro")) < 1e-12
    assert abs(m["jaccard"] - jaccard_score(yt, yp, average="macro")) < 1e-12


def test_metrics_edge_cases():
    assert O.confusion_matrix(np.array([]), np.array([]), 3, None).sum() == 0
    m = O.confusion_metrics(np.zeros((2, 2), dtype=np.int64))
    assert np.isnan(m["accuracy"]) and m["jaccard"] == 0.0  # _safe_div returns 0 where the denominator is 0
    cm = O.confusion_matrix(np.array([-1, -1, 0]), np.array([1, 0, 0]), 2, -1)
    assert cm.tolist() == [[1, 0], [0, 0]]


def test_loss_semantics_not_torch_weighted_mean():
    """segmentation.py:120-122: sum(w_y nll)/#valid, which differs from CrossEntropyLoss(reduction='mean')."""
    g = torch.Generator().manual_seed(0)
    logits = torch.randn(2, 3, 5, 5, generator=g)
    labels = torch.randint(0, 3, (2, 5, 5), generator=g)
    labels[0, 0, :] = -1
    w = torch.tensor([1.0, 3.0, 0.7])
    ours = O.seg_loss(logits, labels, w, -1)
    lsm = torch.log_softmax(logits, 1)
    mask = labels != -1
    nll = -lsm.gather(1, labels.clamp(min=0).unsqueeze(1)).squeeze(1)
    manual = (w[labels.clamp(min=0)] * nll)[mask].sum() / mask.sum()
    assert torch.allclose(ours, manual)
    torch_mean = torch.nn.functional.cross_entropy(logits, labels, weight=w, ignore_index=-1)
    assert not torch.allclose(ours, torch_mean)


def test_window_origins_and_normalize():
    g = np.load(os.path.join(GOLD, "windows.npz"))
    assert np.array_equal(np.array(O.window_origins(512, 224, 224)), g["S512"]) and len(g["S512"]) == 4
    w = O.window_origins(10980, 224, 224)
    assert np.array_equal(np.array(w), g["S10980"]) and len(w) == 49 * 49 and w[-1] == (10752, 10752)
    chip = np.arange(2 * 3 * 4 * 4, dtype=np.float64).reshape(6, 4, 4)  # T=2, C=3, band = t*C + c
    out = O.normalize_chip(chip, [1.0, 2.0, 3.0], [2.0, 4.0, 8.0], 2)
    assert out.shape == (3, 2, 4, 4)
    assert np.allclose(out[1, 1], (chip[1 * 3 + 1] - 2.0) / 4.0)


def test_adamw_and_scheduler_restatement():
    p0, gr = torch.randn(50), torch.randn(50)
    q = torch.nn.Parameter(p0.clone())
    opt = torch.optim.AdamW([q], lr=1e-3, weight_decay=1e-2)
    sch = torch.optim.lr_scheduler.CosineAnnealingWarmRestarts(opt, T_0=10, T_mult=2, eta_min=0)
    p, m, v = p0.clone(), torch.zeros(50), torch.zeros(50)
    for step in range(1, 5):
        q.grad = gr.clone()
        opt.step()
        O.adamw_step(p, gr, m, v, step, lr=1e-3, wd=1e-2)
    assert torch.allclose(p, q.detach(), atol=1e-7)
    for epoch in range(35):
        assert abs(O.cosine_warm_restarts_lr(1e-3, epoch) - opt.param_groups[0]["lr"]) < 1e-12
        sch.step()


def test_state_dict_contract_counts():
    """Parameter counts probed from the reference."""
    for variant, T, ncls, enc, head in [("prithvi_eo_v1_100", 1, 2, 86237184, 5290658), ("prithvi_eo_v1_100", 3, 13, 86237184 + 2 * 196 * 0, 47599645)]:
        cfg = O.make_config(variant, T, ncls)
        shapes = O.state_dict_shapes(cfg)
        n_enc = sum(int(np.prod(s)) for k, s in shapes.items() if k.startswith("prithvi_encoder.") and not k.endswith("pos_embed"))
        n_head = sum(int(np.prod(s)) for k, s in shapes.items() if k.startswith("segmentation_head.") and "running" not in k and "num_batches" not in k)
        assert n_enc == enc and n_head == head


def test_compute_stats_oracle_matches_reference_fixture():
    """mode=stats: the oracle restatement against the outputs of the reference's own compute_stats (gen_golden.py step 5)."""
    from oracle.cases import make_stats_batches

    z = np.load(os.path.join(GOLD, "stats.npz"))
    for case in ("t1", "t3"):
        mean, std, cw = O.compute_stats(make_stats_batches(case))
        assert np.allclose(mean, z[f"{case}_mean"], rtol=2e-6, atol=2e-6)
        assert np.allclose(std, z[f"{case}_std"], rtol=2e-6, atol=2e-6)
        assert np.allclose(cw, z[f"{case}_class_weights"], rtol=1e-12)
    # std is the root of the MEAN per-chip variance, not the pooled standard deviation
    a = torch.zeros(1, 1, 1, 4, 4)
    b = torch.ones(1, 1, 1, 4, 4) * 2
    mean, std, _ = O.compute_stats([(torch.cat([a, b]), torch.zeros(2, 4, 4))])
    assert mean == [1.0] and std == [0.0]
    assert O.compute_class_weights({0: 30, 2: 10}) == [40 / (2 * 30), 0.0, 40 / (2 * 10)]


def test_crop_flip_restatement():
    x = np.arange(2 * 6 * 8).reshape(2, 6, 8)
    y = np.arange(6 * 8).reshape(6, 8) + 100
    cx, cy = O.crop_flip_chip(x, y, top=1, left=2, hflip=False, vflip=False, im_size=4)
    assert np.array_equal(cx, x[:, 1:5, 2:6]) and np.array_equal(cy, y[1:5, 2:6])
    hx, hy = O.crop_flip_chip(x, y, 1, 2, True, False, 4)
    assert np.array_equal(hx[:, :, 0], x[:, 1:5, 5]) and np.array_equal(hy[:, 0], y[1:5, 5])
    vx, vy = O.crop_flip_chip(x, y, 1, 2, True, True, 4)
    assert vx[0, 0, 0] == x[0, 4, 5] and vy[0, 0] == y[4, 5] and vx[1, 3, 3] == x[1, 1, 2]


def test_running_auc_oracle_known_answers_and_fixture():
    """RunningAUC restatement: the reference test-suite's known answers (tests/model_tests/test_metrics.py:77-139: 0.75 and
    0.375 against scikit-learn) and the histograms/scores the reference class produced for a seeded 3-class set."""
    y_true = np.array([1, 1, 1, 1, 1, 0, 1, 1, 0, 1])
    p1 = np.array([0.9, 0.8, 0.7, 0.6, 0.55, 0.45, 0.4, 0.3, 0.2, 0.1])
    pos, neg = O.auc_histograms(y_true, np.stack([1 - p1, p1], 1), 2, n_bins=2048)
    assert O.auc_score(pos, neg)[0] == 0.75
    p2 = np.array([0.6, 0.55, 0.5, 0.45, 0.4, 0.6, 0.55, 0.5, 0.45, 0.4])
    pos, neg = O.auc_histograms(y_true, np.stack([1 - p2, p2], 1), 2, n_bins=2048)
    assert O.auc_score(pos, neg)[0] == 0.375
    z = np.load(os.path.join(GOLD, "auc.npz"))
    pos, neg = O.auc_histograms(z["y_true"], z["probs"], 3, 1024)
    assert np.array_equal(pos, z["pos_hist"]) and np.array_equal(neg, z["neg_hist"])
    macro, per = O.auc_score(pos, neg)
    assert macro == float(z["macro"]) and per == z["per_class"].tolist()
    # a class without positives scores NaN and is left out of the macro mean
    pos, neg = O.auc_histograms(np.array([0, 0, 1]), np.array([[0.7, 0.2, 0.1], [0.6, 0.3, 0.1], [0.2, 0.7, 0.1]]), 3, 64)
    macro, per = O.auc_score(pos, neg)
    assert np.isnan(per[2]) and macro == 1.0


def test_regression_oracle_matches_reference_fixture():
    """Regression metrics restatement against the reference RunningRegressionMetrics outputs (gen_golden.py step 3c) and
    the loss semantics of regression.py:153-168."""
    z = np.load(os.path.join(GOLD, "regression.npz"))
    m = O.regression_metrics(O.regression_sums(z["y_true"], z["y_pred"]), include_ee=True)
    got = np.array([m[k] for k in ("mae", "rmse", "r2_score", "pearson_corrcoef", "ee_percentage")])
    assert np.allclose(got, z["metrics"], rtol=1e-12, atol=0)
    out = torch.tensor([[[[1.0, 2.0], [3.0, 4.0]]]])
    lab = torch.tensor([[[0.0, -100.0], [5.0, 4.0]]])
    loss, preds, l2 = O.regression_loss(out, lab, -100.0)
    assert abs(loss.item() - (1.0 + 4.0 + 0.0) / 3) < 1e-6 and preds.tolist() == [1.0, 3.0, 4.0] and l2.tolist() == [0.0, 5.0, 4.0]
    loss2, preds2, l22 = O.regression_loss(out, lab, -100.0, use_log_scale=True)
    ref = ((torch.tensor([1.0, 3.0, 4.0]) - torch.log1p(torch.tensor([0.0, 5.0, 4.0]))) ** 2).mean()
    assert torch.allclose(loss2, ref) and torch.allclose(preds2, torch.expm1(torch.tensor([1.0, 3.0, 4.0])))
    e = O.regression_metrics([1, 1.0, 1.0, 1.0, 1.0, 1.0, 0.0, 0.0, 1.0])
    assert np.isnan(e["r2_score"]) and np.isnan(e["pearson_corrcoef"]) and e["mae"] == 0.0


def test_distillation_loss_restatement():
    """CE + KLDiv(batchmean) of segmentation.py:352-378 on a hand-checkable case."""
    s_log = torch.tensor([[[[0.0, 2.0]], [[0.0, 0.0]]]])  # (1, 2, 1, 2): pixel 0 logits (0,0), pixel 1 logits (2,0)
    t_log = torch.tensor([[[[0.0, 0.0]], [[0.0, 2.0]]]])  # teacher: pixel 0 (0,0), pixel 1 (0,2)
    lab = torch.tensor([[[1.0, -1.0]]])                   # pixel 1 ignored
    total, ce, kd = O.distillation_loss(s_log, t_log, lab, -1)
    assert abs(ce.item() - np.log(2.0)) < 1e-6 and abs(kd.item()) < 1e-7  # same distribution on the only valid pixel
    lab2 = torch.tensor([[[1.0, 0.0]]])
    total, ce, kd = O.distillation_loss(s_log.double(), t_log.double(), lab2, -1)
    p = np.exp(2) / (1 + np.exp(2))
    kl_pix1 = (1 - p) * (np.log(1 - p) - np.log(p)) + p * (np.log(p) - np.log(1 - p))  # t = (1-p, p), s = (p, 1-p)
    assert abs(kd.item() - kl_pix1 / 2) < 1e-12 and abs(total.item() - (ce + kd).item()) < 1e-15


def test_task_losses_oracle_matches_reference_generated_fixture():
    """tests/golden/task_losses.npz holds the loss parts and autograd gradients of the reference's own
    PrithviDistillationSegmentationModule._compute_loss (segmentation.py:352-378), PrithviRegressionModule._shared_step
    (regression.py:141-168) and PrithviDistillationRegressionModule._shared_step / _compute_loss (regression.py:477-534), written by
    oracle/gen_golden.py from an import of /root/reference.  The oracle restatements must reproduce them in fp64."""
    z = np.load(os.path.join(GOLD, "task_losses.npz"))
    s = torch.from_numpy(z["seg_student"]).double().requires_grad_(True)
    total, ce, kd = O.distillation_loss(s, torch.from_numpy(z["seg_teacher"]).double(), torch.from_numpy(z["seg_labels"]), int(z["seg_ignore"]),
                                        torch.from_numpy(z["seg_class_weights"]).double())
    total.backward()
    assert np.allclose([total.item(), ce.item(), kd.item()], z["seg_parts_f64"], rtol=0, atol=1e-12)
    assert np.abs(s.grad.numpy() - z["seg_grad_f64"]).max() < 1e-14
    assert np.abs(z["seg_parts_f32"] - z["seg_parts_f64"]).max() < 1e-5  # the reference's own fp32 run of the same method
    lab = torch.from_numpy(z["reg_labels"]).double()
    ign = float(z["reg_ignore"])
    for use_log, key in ((False, "lin"), (True, "log")):
        s = torch.from_numpy(z["reg_student"]).double().requires_grad_(True)
        loss, preds, l2 = O.regression_loss(s, lab, ign, use_log)
        loss.backward()
        assert abs(loss.item() - float(z[f"reg_loss_{key}_f64"])) < 1e-12 and np.abs(s.grad.numpy() - z[f"reg_grad_{key}_f64"]).max() < 1e-14
        m = O.regression_metrics(O.regression_sums(l2.numpy(), preds.numpy()), include_ee=True)
        got = np.array([m[k] for k in ("mae", "rmse", "r2_score", "pearson_corrcoef", "ee_percentage")])
        assert np.allclose(got, z[f"reg_metrics_{key}_f64"], rtol=1e-12, atol=1e-12)  # the step's metrics.update(labels, preds) operands
        s = torch.from_numpy(z["reg_student"]).double().requires_grad_(True)
        tot, mse, kdl = O.regression_distillation_loss(s, torch.from_numpy(z["reg_teacher"]).double(), lab, ign, use_log)
        tot.backward()
        assert np.allclose([tot.item(), mse.item(), kdl.item()], z[f"regkd_parts_{key}_f64"], rtol=0, atol=1e-12)
        assert np.abs(s.grad.numpy() - z[f"regkd_grad_{key}_f64"]).max() < 1e-14


def test_interpolated_pos_embed_matches_reference_fixture():
    """interpolate_pos_encoding (pritvhi.py:149-203; inputs off the configured 224 grid): the oracle restatement against the rows
    gen_golden.py stored from the imported reference function (bicubic, align_corners=True, cls row kept)."""
    gold = np.load(os.path.join(GOLD, "pos_embed.npz"))
    for T in (1, 3):
        cfg = O.make_config("prithvi_eo_tiny", T, 2, 224)
        table = torch.from_numpy(O.sincos_pos_embed_3d(256, (T, 14, 14), True)[None].astype(np.float32))
        assert O.interpolate_pos_encoding(cfg, table, 224, 224) is table
        for S in (160, 256):
            pe = O.interpolate_pos_encoding(cfg, table, S, S)
            g = S // 16
            assert pe.shape == (1, 1 + T * g * g, 256)
            rows = pe[0, [0, 1, 2, g * g, pe.shape[1] - 1]].numpy()
            assert np.allclose(rows, gold[f"interp_T{T}_S{S}_rows"], rtol=0, atol=1e-6)
            assert abs(pe.double().sum().item() - float(gold[f"interp_T{T}_S{S}_sum"])) <= 1e-3
            assert torch.all(pe[0, 0] == 0)  # the cls row of the sin-cos table is zeros and is not interpolated
