"""Training losses: counterpart of the reference's `loss.py` (BASELINE config 4).

`Loss(config)(gt, pred)` with the reference's config keys: `BPPLoss` (`loss.py:63-81`), `ColorLoss` (`:84-111`),
`Multiscale_FocalLoss` (`:115-157`).  `ShepardsLoss` (ablation, `:161-274`) is out of scope.  Coordinate membership
tests use exact packed keys on the device (the reference flattens coordinates with float-scaled weights and torch.isin).
"""
import math
import os

import torch

from . import lib as L
from . import sparse as S

FUSED_FOCAL = os.environ.get("PCC_FUSED_FOCAL", "1") != "0"      # focal loss rows as one kernel per level (training step)


def _lookup_rows32(cset, query_keys, nq):
    rows = torch.empty(max(nq, 1), dtype=torch.int32, device=cset.device)
    if nq:
        L.call("pcc_lookup_rows", L.ptr(cset.keys), cset.n, L.ptr(query_keys), nq, L.ptr(rows), L.stream())
    return rows[:nq]


def _lookup_rows(cset, query_keys, nq):
    return _lookup_rows32(cset, query_keys, nq).long()


class BPPLoss:
    def __init__(self, config):
        self.weight, self.identifier, self.key = config["weight"], config["id"], config["key"]

    def __call__(self, gt, pred):
        lik = pred["likelihoods"][self.key]
        return (torch.log(lik).sum() / (-math.log(2) * gt.C.shape[0])) * self.weight


class ColorLoss:
    def __init__(self, config):
        self.identifier = config["id"]
        self.l2 = config["loss"] == "L2"

    def __call__(self, gt, pred):
        prediction, q_map = pred["prediction"], pred["q_map"]
        gcs, pcs = gt._cset, prediction._cset
        # The overlapping voxels are enumerated from the PREDICTION side: the ground-truth row of every decoded voxel (-1: none),
        # the ground-truth colours gathered (no gradient), the prediction's own rows used in place -- so the backward pass has no
        # scatter in it.  A 0/1 weight replaces boolean-mask indexing (which makes torch count the mask on the host, in both
        # directions).  Same pairs and weights as `loss.py:84-111`; mean over overlapping voxels x channels.
        rows = _lookup_rows(gcs, pcs.keys, pcs.n)
        ov = (rows >= 0).to(torch.float32).unsqueeze(1)
        gt_colors = gt._canonical_features().index_select(0, rows.clamp(min=0))
        pred_colors = prediction._canonical_features()
        batch = pcs.keys[:pcs.n] >> 48
        d = (gt_colors - pred_colors) * ov
        e = d * d if self.l2 else d.abs()
        return (e * q_map[batch, 1].unsqueeze(1)).sum() / (ov.sum() * gt_colors.shape[1])


class Multiscale_FocalLoss:
    def __init__(self, config):
        self.identifier, self.alpha, self.gamma = config["id"], config["alpha"], config["gamma"]

    def __call__(self, gt, pred):
        predictions, points, q_map = list(pred["occ_predictions"]), list(pred["points"]), pred["q_map"]
        predictions.reverse()
        points.reverse()
        loss = 0.0
        for prediction, coords in zip(predictions, points):
            pcs, gcs = prediction._cset, coords._cset
            logit = prediction._canonical_features()[:, 0]
            if FUSED_FOCAL and logit.is_cuda and pcs.n > 0:        # one kernel + one sum per level (`autograd.FocalRowsFn`)
                from .autograd import FocalRowsFn
                occ_row = _lookup_rows32(gcs, pcs.keys, pcs.n)
                loss = loss + FocalRowsFn.apply(logit, occ_row, pcs.keys, q_map, self.alpha, self.gamma) / pcs.n
                continue
            occ = _lookup_rows(gcs, pcs.keys, pcs.n) >= 0           # predicted voxel is occupied in the ground truth
            p = torch.sigmoid(prediction._canonical_features()[:, 0])
            pt = torch.clip(torch.where(occ, p, 1 - p), 1e-2, 1)
            alpha = torch.where(occ, self.alpha, 1 - self.alpha)
            focal = -alpha * (1 - pt) ** self.gamma * torch.log(pt)
            batch = pcs.keys[:pcs.n] >> 48
            loss = loss + (focal * q_map[batch, 0]).mean()
        return loss


class Loss:
    def __init__(self, config):
        self.losses = {}
        for ident, setting in config.items():
            setting = dict(setting, id=ident)
            kind = setting["type"]
            if kind == "BPPLoss":
                self.losses[ident] = BPPLoss(setting)
            elif kind == "ColorLoss":
                self.losses[ident] = ColorLoss(setting)
            elif kind == "Multiscale_FocalLoss":
                self.losses[ident] = Multiscale_FocalLoss(setting)
            else:
                raise L.PccError(f"loss {kind!r} is out of scope (ablation only)")

    def __call__(self, gt, pred):
        total, parts = 0, {}
        for loss in self.losses.values():
            item = loss(gt, pred)
            parts[loss.identifier] = item
            total = total + item
        return total, parts
