"""TEST INFRASTRUCTURE ONLY (see oracle/ugaitnet_oracle.py).  CPU restatement of the batch assembly of the reference's
three-modality generator, data/mj_dataGeneratorMMUWYHsingle_repetitions.py: `__load_dd` (:276-329) and `__gen_batchMM`
(:658-818) with augmentation off, gaitset=True, ntype=2.  PARITY UNPINNED beyond the code reading: the reference's own
generator needs deepdish + TensorFlow, neither installed here; there are no fixtures for it in the reference's tree."""
import random

import numpy as np

NOISE = 0.000000001   # :102


def load_dd(data, compress_factor, silhouette=False, ntype=2, clip_max=0, clip_min=0):
    """`data` as stored: int16 [60,60,50] (compressFactor > 1) or uint8 [60,60,25] -> float32 [T,60,60]  (:300-327)."""
    if compress_factor > 1:
        x = np.float32(data)
        if clip_max > 0:
            x[np.abs(x) > clip_max] = 1e-8
        if clip_min > 0:
            x[np.abs(x) < clip_min] = 1e-8
        x = x / np.float32(compress_factor)          # float32 array / scalar: stays float32 (numpy 1.x value-based casting)
        if ntype == 2:
            x = x * np.float32(0.1)
    else:
        if silhouette:
            x = np.float32(data) / np.float32(255.0)
        else:
            x = (np.float32(data) / np.float32(255.0)) - np.float32(0.5)
    if ntype == 2:
        x = np.moveaxis(x, 2, 0)                       # :321-323
    return x


def gaitset_layout(x_tmp):
    """[50|25,60,60] -> [25,60,60,2|1]  (:746-753)."""
    if x_tmp.shape[0] == 50:
        x_new = np.zeros((25, x_tmp.shape[1], x_tmp.shape[2], 2), dtype=x_tmp.dtype)
        x_new[:, :, :, 0] = x_tmp[::2, :, :]
        x_new[:, :, :, 1] = x_tmp[1::2, :, :]
    else:
        x_new = np.zeros((25, x_tmp.shape[1], x_tmp.shape[2], 1), dtype=x_tmp.dtype)
        x_new[:, :, :, 0] = x_tmp
    return x_new


def gen_batch_mm(samples, specs, expand, seed, clip=(0, 0)):
    """samples[i][j]: raw `data` array of base sample i, modality j, or None (file absent).
    specs[j] = dict(compress_factor=..., silhouette=bool, channels=1|2).  Returns x list [x0,u0,x1,u1,...] as float32
    (the reference's float64 arrays hold float32 values; Keras casts them back) and the row plan actually drawn."""
    rng = random.Random(seed)
    nbase, nmods = len(samples), len(specs)
    expand = max(1, expand)
    dim0 = nbase * expand
    x = []
    for sp in specs:
        x += [np.empty((dim0, 25, 60, 60, sp["channels"])), np.empty((dim0, 1))]
    plan = np.full((dim0, nmods), -1, np.int32)
    for i in range(nbase):
        for mix in range(nmods):
            d = samples[i][mix]
            if d is None:                                                   # :735-737
                x[2 * mix][i * expand,] = NOISE
                x[2 * mix + 1][i * expand,] = 0.0
            else:
                sp = specs[mix]
                cm, cn = clip if sp["compress_factor"] > 1 else (0, 0)
                x_tmp = gaitset_layout(load_dd(d, sp["compress_factor"], sp.get("silhouette", False), 2, cm, cn))
                x[2 * mix][i * expand,] = x_tmp                              # :755-756
                x[2 * mix + 1][i * expand,] = 1.0
                plan[i * expand, mix] = i
        if expand > 1:                                                       # :776-806
            nmore = expand - 1
            for ex in range(nmore):
                if i % 2 == 0:
                    if expand > 2:
                        ndisable = min(ex + 1, nmods - 1)
                    else:
                        ndisable = rng.randrange(1, nmods, 1)
                    l_dis = [1] * nmods
                    for ff in range(ndisable):
                        choice1 = rng.randrange(0, nmods, 1)
                        l_dis[choice1] = 0
                else:
                    l_dis = [0] * nmods
                    l_dis[(i + ex) % 3] = 1
                for j in range(nmods):
                    if l_dis[j] == 0:
                        x[2 * j][(ex + 1) + i * expand,] = NOISE
                        x[2 * j + 1][(ex + 1) + i * expand,] = 0.0
                    else:
                        x[2 * j][(ex + 1) + i * expand,] = np.copy(x[2 * j][i * expand,])
                        x[2 * j + 1][(ex + 1) + i * expand,] = 1.0
                        plan[(ex + 1) + i * expand, j] = i
    return [a.astype(np.float32) for a in x], plan


def gen_batch_2mod(samples, specs, expand, seed):
    """The two-modality generator `__gen_batch` (:347-545) with augmentation off, gaitset=True: row 0 of a sample carries what
    exists; row 1 disables ONE randomly chosen modality and copies the other (flag 1 even when the copy is noise); row 2
    (expand > 2) swaps the choice.  expand is 1, 2 or 3 (the reference fills no further rows)."""
    rng = random.Random(seed)
    nbase = len(samples)
    expand = max(1, min(expand, 3))
    dim0 = nbase * expand
    x = []
    for sp in specs:
        x += [np.empty((dim0, 25, 60, 60, sp["channels"])), np.empty((dim0, 1))]
    plan = np.full((dim0, 2), -1, np.int32)
    for i in range(nbase):
        for mix in range(2):
            d = samples[i][mix]
            if d is None:                                                   # :414-415, 439-441, 471-472
                x[2 * mix][i * expand,] = NOISE
                x[2 * mix + 1][i * expand,] = 0.0
            else:
                sp = specs[mix]
                x[2 * mix][i * expand,] = gaitset_layout(load_dd(d, sp["compress_factor"], sp.get("silhouette", False), 2))
                x[2 * mix + 1][i * expand,] = 1.0
                plan[i * expand, mix] = i
        if expand > 1:                                                       # :485-528
            choice = rng.randrange(0, 2, 1)
            for row in range(1, expand):
                off, on = choice, 1 - choice
                x[2 * off][row + i * expand,] = NOISE
                x[2 * off + 1][row + i * expand,] = 0.0
                x[2 * on][row + i * expand,] = np.copy(x[2 * on][i * expand,])
                x[2 * on + 1][row + i * expand,] = 1.0
                plan[row + i * expand, on] = i
                choice = 1 - choice
    return [a.astype(np.float32) for a in x], plan
