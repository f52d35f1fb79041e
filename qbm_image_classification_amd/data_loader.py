"""Dataset readers with the reference's output contract (boundary B4): same function names,
arguments, shapes, dtypes and ordering as src/data_loader.py, so that the reference drivers' data plumbing
runs unchanged -- plus the build-side pieces the reference lacks: binarisation, uint8 / bit packing for the
device-resident pipeline, and a Bars-and-Stripes generator (BASELINE.json configs[0])."""
from __future__ import annotations

import gzip

import numpy as np


def get_mnist(file_image, file_labels, classes=None, size=None, samples_per_class=None, seed=42):
    """src/data_loader.py:23-73 -> (float32 [N,28,28] in [0,1], labels: list for 2 classes else uint8 array)"""
    with gzip.open(file_image, 'rb') as imgpath:
        images = np.frombuffer(imgpath.read(), 'B', offset=16).reshape(-1, 28, 28).astype('float32') / 255
    with gzip.open(file_labels, 'rb') as lbpath:
        labels = np.frombuffer(lbpath.read(), 'B', offset=8)
    return _select(images, labels, classes, size, samples_per_class, seed)


def get_fashionmnist(file_image, file_labels, classes=None, size=None, samples_per_class=None, seed=42):
    """src/data_loader.py:76-112 (uncompressed idx files)"""
    with open(file_image, 'rb') as imgpath:
        images = np.frombuffer(imgpath.read(), 'B', offset=16).reshape(-1, 28, 28).astype('float32') / 255
    with open(file_labels, 'rb') as lbpath:
        labels = np.frombuffer(lbpath.read(), 'B', offset=8)
    return _select(images, labels, classes, size, samples_per_class, seed)


def _select(images, labels, classes, size, samples_per_class, seed):
    if classes is not None:
        mask = np.isin(labels, classes)
        images, labels = images[mask], labels[mask]
    if samples_per_class is not None:
        sel_i, sel_l = [], []
        np.random.seed(seed)
        for cls in np.unique(labels):
            idx = np.random.choice(np.where(labels == cls)[0], size=samples_per_class, replace=False)
            sel_i.append(images[idx]); sel_l.append(labels[idx])
        images, labels = np.concatenate(sel_i), np.concatenate(sel_l)
    if size is not None:
        from sklearn.model_selection import train_test_split

        images, _, labels, _ = train_test_split(images, labels, train_size=size, random_state=seed)
    if classes is not None and len(classes) == 2:
        labels = [0 if label == classes[0] else 1 for label in labels]   # a python list, like the reference (:68-69)
    print("Images: ", len(images))
    return images, labels


def get_medmnist(file, index=0, duplicate_positives_n_times=0, do_balance=False, size=None, seed=42):
    """src/data_loader.py:115-167 -> ((train_x, train_y), (val_x, val_y), (test_x, test_y)), float32 / uint8.
    Like the reference this reseeds numpy's global RNG (:121)."""
    data = np.load(file)
    np.random.seed(seed)
    train_images = data['train_images'].astype('float32') / 255
    val_images = data['val_images'].astype('float32') / 255
    test_images = data['test_images'].astype('float32') / 255
    train_labels = data['train_labels'][:, index]
    val_labels = data['val_labels'][:, index]
    test_labels = data['test_labels'][:, index]
    if size is not None:
        sel_i, sel_l = [], []
        for cls, cls_size in enumerate(size):
            idx = np.where(train_labels == cls)[0]
            idx = np.random.choice(idx, size=cls_size, replace=len(idx) <= cls_size)
            sel_i.append(train_images[idx]); sel_l.append(train_labels[idx])
        train_images, train_labels = np.concatenate(sel_i), np.concatenate(sel_l)
    if duplicate_positives_n_times > 0:
        pos = np.where(train_labels == 1)[0]
        for _ in range(duplicate_positives_n_times):
            train_images = np.concatenate([train_images, train_images[pos]])
            train_labels = np.concatenate([train_labels, train_labels[pos]])
    if do_balance:
        pos = np.where(train_labels == 1)[0]
        neg = np.random.choice(np.where(train_labels == 0)[0], len(pos))
        train_images = np.concatenate([train_images[pos], train_images[neg]])
        train_labels = np.concatenate([train_labels[pos], train_labels[neg]])
    return (train_images, train_labels), (val_images, val_labels), (test_images, test_labels)


def resize_and_flatten(train_x, test_x, val_x=None, outputshape=None):
    """src/data_loader.py:227-237"""
    tr = np.array([im.flatten() for im in train_x])
    te = np.array([im.flatten() for im in test_x])
    va = np.array([im.flatten() for im in val_x]) if val_x is not None else None
    return tr, te, va


def preprocess_images(train_x, test_x, val_x=None, outputshape=None, pca_n_components=None, do_flatten=True):
    """src/data_loader.py:240-276 -> (train, test, val) flattened [N,784]; note the positional order."""
    print("Start preprocessing images for encoding...")
    if do_flatten:
        train_x, test_x, val_x = resize_and_flatten(train_x, test_x, val_x, outputshape)
    if pca_n_components is not None:
        from sklearn.decomposition import PCA

        pca = PCA(n_components=pca_n_components)
        train_x = pca.fit_transform(train_x)
        test_x = pca.transform(test_x)
        val_x = pca.transform(val_x)
    print("Preprocessing images for encoding finished!!")
    return train_x, test_x, val_x,


def shuffle_images(x, y, seed=44):
    """src/data_loader.py:279-285"""
    assert len(x) == len(y), "Input data and labels must have the same length."
    idx = np.random.default_rng(seed).permutation(len(x))
    return np.take(x, idx, axis=0), np.take(y, idx, axis=0)


def encode_labels_to_onehot(labels, num_classes):
    """src/data_loader.py:288-296"""
    return np.eye(num_classes)[labels]


# ---- build-side additions (not in the reference) --------------------------------------------------
def binarise(x, threshold=0.5):
    """{0,1} uint8 states from greyscale in [0,1]: x > threshold (the threshold the fixtures use)."""
    return (np.asarray(x) > threshold).astype(np.uint8)


def pack_bits(x_bin):
    """[N, V] {0,1} -> [N, ceil(V/8)] uint8, MSB first (numpy.packbits); 98 bytes per 28x28 image."""
    return np.packbits(np.asarray(x_bin, dtype=np.uint8), axis=1)


def unpack_bits(bits, num_visible):
    return np.unpackbits(np.asarray(bits, dtype=np.uint8), axis=1)[:, :num_visible]


def bars_and_stripes(side=4, include_trivial=False):
    """All Bars-and-Stripes patterns of a side x side grid -> (x uint8 [N, side*side], y int64: 0 = bars
    (constant columns), 1 = stripes (constant rows)).  The two constant images are excluded by default
    (2^(side+1) - 4 = 28 patterns for side 4... 30 with one copy of each trivial image kept per class)."""
    xs, ys = [], []
    for bits in range(1 << side):
        v = np.array([(bits >> i) & 1 for i in range(side)], dtype=np.uint8)
        trivial = bits == 0 or bits == (1 << side) - 1
        if trivial and not include_trivial:
            continue
        xs.append(np.tile(v, (side, 1)).reshape(-1)); ys.append(0)            # bars: every row equals v
        xs.append(np.tile(v[:, None], (1, side)).reshape(-1)); ys.append(1)   # stripes: every column equals v
    return np.stack(xs), np.array(ys, dtype=np.int64)
