"""The two names the reference's callers import from its utils module (reference test_last.py:13, train.py):
`setup_seed` and `cos_sim` (reference utils.py:10-21, :88-95).  Host-side helpers, not part of the hot path; the
reference's augmentation helpers (rot_img, AddGaussianNoise, ...) belong to training and are out of scope
(SURVEY.md section 2)."""
import os
import random

import numpy as np
import torch


def setup_seed(seed: int) -> None:
    """Seed every generator a caller can reach (reference utils.py:10-21)."""
    torch.manual_seed(seed)
    if torch.cuda.is_available():
        torch.cuda.manual_seed_all(seed)
    np.random.seed(seed)
    random.seed(seed)
    os.environ["PYTHONHASHSEED"] = str(seed)


def cos_sim(a_norm: torch.Tensor, b_norm: torch.Tensor) -> torch.Tensor:
    """Similarity of unit rows b with unit row(s) a (reference utils.py:88-95): b @ a^T for a 2-D a, b @ a for 1-D."""
    if a_norm.dim() == 2:
        return b_norm @ a_norm.transpose(1, 0)
    if a_norm.dim() == 1:
        return b_norm @ a_norm
    raise NotImplementedError
