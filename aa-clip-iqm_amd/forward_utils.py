"""Text anchors and similarity maps with the reference's function names
(reference forward_utils.py:138-216), executed by the HIP path."""
from __future__ import annotations

from typing import Sequence

import torch

from aaclip_hip import engine
from dataset.constants import CLASS_NAMES, DOMAINS, PROMPTS, REAL_NAMES  # noqa: F401  (DOMAINS: re-exported like the reference)
from model.tokenizer import tokenize

prompt_normal = PROMPTS["prompt_normal"]
prompt_abnormal = PROMPTS["prompt_abnormal"]
prompt_state = [prompt_normal, prompt_abnormal]
prompt_templates = PROMPTS["prompt_templates"]


def class_sentences(dataset_name: str, class_name: str):
    """The 6 normal + 10 abnormal sentences of one class (forward_utils.py:139-152)."""
    if class_name == "object":
        real_name = class_name
    else:
        assert class_name in CLASS_NAMES[dataset_name], (
            f"class_name {class_name} not found; available class_names: {CLASS_NAMES[dataset_name]}")
        real_name = REAL_NAMES[dataset_name][class_name]
    return [[tpl.format(state.format(real_name)) for state in states for tpl in prompt_templates]
            for states in prompt_state]


def _anchor(embeddings: torch.Tensor) -> torch.Tensor:
    # [n, E] sentence embeddings -> unit mean of unit rows (forward_utils.py:155-159); n <= 10 rows, host-side math
    e = embeddings / embeddings.norm(dim=-1, keepdim=True)
    m = e.mean(dim=0)
    return m / m.norm()


def get_adapted_single_class_text_embedding(model, dataset_name, class_name, device):
    """-> [E, 2]: column 0 normal, column 1 abnormal (forward_utils.py:138-162)."""
    cols = []
    for sentences in class_sentences(dataset_name, class_name):
        tokens = tokenize(sentences).to(device)
        cols.append(_anchor(model.encode_text(tokens)))
    return torch.stack(cols, dim=1).to(device)


def get_adapted_text_embedding(model, dataset_name, device):
    """dict class -> [E, 2] (forward_utils.py:185-192).  All classes x states go
    through ONE batched encode_text call (the tiny M = n*77 GEMMs are launch-bound
    one class at a time), then a segmented mean."""
    names = list(CLASS_NAMES[dataset_name])
    groups, sentences = [], []
    for c in names:
        for s in class_sentences(dataset_name, c):
            groups.append((c, len(sentences), len(s)))
            sentences.extend(s)
    emb = model.encode_text(tokenize(sentences).to(device))
    out = {}
    for c in names:
        cols = [_anchor(emb[start:start + n]) for (cc, start, n) in groups if cc == c]
        out[c] = torch.stack(cols, dim=1).to(device)
    return out


def calculate_similarity_map(patch_features, epoch_text_feature, img_size, test=False, domain="Medical"):
    """forward_utils.py:196-216.  test=True -> [B,1,S,S] blurred/upsampled anomaly
    map; test=False -> [B,2,S,S] softmax over the two anchors."""
    if test:
        assert epoch_text_feature.shape[-1] == 2
        sigma, ksize = (1.0, 7) if domain == "Industrial" else (1.5, 9)
        return engine.anomaly_map([patch_features], epoch_text_feature, img_size, ksize, sigma).unsqueeze(1)
    return engine.similarity_map_train(patch_features, epoch_text_feature, img_size)


def calculate_anomaly_map(patch_features: Sequence[torch.Tensor], epoch_text_feature, img_size, domain="Industrial"):
    """All tap levels in one pass: sum over levels of calculate_similarity_map(test=True)
    (what test_last.py:95-100,149 computes with 4 calls + cat + sum) -> [B,S,S]."""
    sigma, ksize = (1.0, 7) if domain == "Industrial" else (1.5, 9)
    return engine.anomaly_map(list(patch_features), epoch_text_feature, img_size, ksize, sigma)


def image_score(det_feature: torch.Tensor, text_feature: torch.Tensor) -> torch.Tensor:
    """Per-image anomaly score (det_b . t_abnormal + 1)/2 -> [B].  Deliberate
    deviation from test_last.py:90-91, whose [B,768]@[B,768,2] broadcast yields
    [B,B,2] and then row 1; B dot products, host-side."""
    t = text_feature if text_feature.dim() == 2 else text_feature[0]
    return (det_feature @ t[:, 1].to(det_feature.device) + 1) / 2


# ------------------------------------------------------------------------------------------------
# evaluation harness counterpart (host side: numpy + sklearn, like the reference)
def metrics_eval(pixel_label, image_label, pixel_preds, image_preds, class_names, domain):
    """Pixel / image AUROC and AP of one class, reference forward_utils.py:233-308:
    global min-max normalisation of maps and image scores (:246-253), per-image maximum of
    the normalised map (:277), Industrial score = 0.5*max_pixel + 0.5*image score, Medical =
    max_pixel (:279-282), sklearn roc_auc_score / average_precision_score (:288-296).
    `image_preds` is the per-image score vector [N] (this build computes the intended
    (det . t_abnormal + 1)/2 score; the reference's [N,2] broadcast quirk, of which it keeps
    column 0, is documented in DESIGN.md and accepted here too)."""
    import numpy as np
    from sklearn.metrics import average_precision_score, roc_auc_score

    # arithmetic stays in the callers' dtype (float32 arrays in test_last.py), as in the reference: the min-max
    # normalisation decides which pixels tie, and ties decide AUROC / AP digits
    pixel_preds = np.asarray(pixel_preds)
    image_preds = np.asarray(image_preds)
    pixel_label = np.asarray(pixel_label)
    image_label = np.asarray(image_label)
    if pixel_preds.max() != 1:
        pixel_preds = (pixel_preds - pixel_preds.min()) / (pixel_preds.max() - pixel_preds.min())
    if image_preds.max() != 1:
        image_preds = (image_preds - image_preds.min()) / (image_preds.max() - image_preds.min())
    if pixel_preds.ndim == 4 and pixel_preds.shape[1] == 1:
        pixel_preds = pixel_preds[:, 0]
    elif pixel_preds.ndim == 2:                      # [N, pixels] -> [N, side, side] (:261-267)
        side = int(pixel_preds.shape[1] ** 0.5)
        if side * side == pixel_preds.shape[1]:
            pixel_preds = pixel_preds.reshape(pixel_preds.shape[0], side, side)
    if image_preds.ndim == 2 and image_preds.shape[1] == 2:
        image_preds = image_preds[:, 0]
    elif image_preds.ndim > 1:
        image_preds = image_preds.reshape(-1)
    pmax = pixel_preds.max(axis=(1, 2))
    image_preds = pmax * 0.5 + image_preds * 0.5 if domain != "Medical" else pmax
    pl, pp = pixel_label.reshape(-1), pixel_preds.reshape(-1)
    pixel_auc = roc_auc_score(pl, pp)
    pixel_ap = average_precision_score(pl, pp)
    if image_label.max() != image_label.min():
        il = image_label.reshape(-1)
        image_auc = roc_auc_score(il, image_preds.reshape(-1))
        image_ap = average_precision_score(il, image_preds.reshape(-1))
    else:
        image_auc = image_ap = 0
    return {"class name": class_names, "pixel AUC": round(pixel_auc, 4) * 100, "pixel AP": round(pixel_ap, 4) * 100,
            "image AUC": round(image_auc, 4) * 100, "image AP": round(image_ap, 4) * 100}


def visualize(*args, **kwargs):
    """The reference's heat-map writer (forward_utils.py:316-360: cv2 colour maps + file output) is outside this
    build (OUT OF SCOPE).  The name exists so that `from forward_utils import visualize`
    (reference test_last.py:17-22) resolves; calling it -- `--visualize` -- says so instead of failing on cv2."""
    raise NotImplementedError("forward_utils.visualize: visualisation is outside the MI355X hot-path build; take the "
                              "[N, S, S] maps test_last.get_predictions returns and plot them with your own tooling")
