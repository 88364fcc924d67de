"""Host helpers the driver imports from ``sac_cbf_clf.utils``
(reference: ``U/sac_cbf_clf/utils.py:60-84,107-141``)."""
import os

import numpy as np
import torch


def prGreen(prt): print("\033[92m {}\033[00m".format(prt))


def prYellow(prt): print("\033[93m {}\033[00m".format(prt))


def prRed(prt): print("\033[91m {}\033[00m".format(prt))


def to_numpy(x):
    return x.detach().cpu().double().numpy()


def to_tensor(x, dtype, device, requires_grad=False):
    return torch.from_numpy(np.asarray(x)).type(dtype).to(device).requires_grad_(requires_grad)


def soft_update(target, source, tau):
    """Polyak step for module pairs living outside the fused device path."""
    with torch.no_grad():
        for t, s in zip(target.parameters(), source.parameters()):
            t.mul_(1.0 - tau).add_(s, alpha=tau)


def hard_update(target, source):
    with torch.no_grad():
        for t, s in zip(target.parameters(), source.parameters()):
            t.copy_(s)


def get_output_folder(parent_dir, env_name):
    """``<parent_dir>/<env_name>-run<N+1>`` with N the highest existing run id."""
    os.makedirs(parent_dir, exist_ok=True)
    ids = [0]
    for name in os.listdir(parent_dir):
        if os.path.isdir(os.path.join(parent_dir, name)) and '-run' in name:
            tail = name.split('-run')[-1]
            if tail.isdigit():
                ids.append(int(tail))
    out = os.path.join(parent_dir, '{}-run{}'.format(env_name, max(ids) + 1))
    os.makedirs(out, exist_ok=True)
    return out
