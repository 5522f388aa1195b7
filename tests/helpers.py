import os

import numpy as np
import torch

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load_golden(name):
    return np.load(os.path.join(GOLDEN, name), allow_pickle=False)


def t(a):
    return torch.from_numpy(np.ascontiguousarray(a))


def rel_err(a, b):
    a = torch.as_tensor(a).double().cpu()
    b = torch.as_tensor(b).double().cpu()
    return ((a - b).abs().max() / (b.abs().max() + 1e-30)).item()


def golden_params(z):
    return {k[len("param."):]: t(z[k]) for k in z.files if k.startswith("param.")}


def golden_case(z, cname):
    pre = "case.%s." % cname
    return {k[len(pre):]: z[k] for k in z.files if k.startswith(pre)}
