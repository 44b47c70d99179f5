"""Development aid: error of the ragged-epoch fixture (3 Adam steps, all weights trainable) under the A/B switches."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, pandas as pd, torch
from conftest import Golden, build_model, pkg
from oracle import ynet_oracle as O
dev = torch.device("cuda:0")
for case in sys.argv[1:] or ["tiny_short_train"]:
    g = Golden(case); cfg, m = g.cfg(), g.meta
    model = build_model(cfg, g.state_dict(), dev)
    te, trn = pkg("utils.train_epoch"), pkg("models.trainer")
    S = cfg.template_size
    in_t, gt_t = O.dist_template(S).to(dev), O.gaussian_template(S, cfg.kernlen, cfg.nsig).to(dev)
    opt = torch.optim.Adam(model.parameters(), lr=m["lr"])
    traj = g.t("epoch/traj")
    loader = [(traj.clone(), [pd.DataFrame({"metaId": np.arange(traj.shape[0])})], "scene0")]
    ade, fde, loss = te.train_epoch(model, loader, {"scene0": g.t("scene")[0]}, opt, trn.HipBCEWithLogitsLoss(), cfg.loss_scale, dev,
                                    "sdd", None, gt_t, in_t, list(cfg.waypoints), 0, cfg.obs_len, cfg.pred_len, m["B"], 10000,
                                    cfg.resize_factor, cfg.network, False)
    print(case, "GRAPH", os.environ.get("YNET_STEP_GRAPH", "1"), "PRED_BCE", os.environ.get("YNET_PRED_BCE", "1"), "FUSED_ADAM", os.environ.get("YNET_FUSED_ADAM", "1"),
          "dADE %.3e dFDE %.3e dloss_rel %.3e" % (ade - float(g.z["epoch/ade"]), fde - float(g.z["epoch/fde"]), (loss - float(g.z["epoch/loss"])) / float(g.z["epoch/loss"])), "lr", m["lr"], "B", m["B"])
