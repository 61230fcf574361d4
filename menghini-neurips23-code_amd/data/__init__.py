"""Dataset objects with the attribute surface the hot path touches (`data/dataset.py:12-89` of the
reference: filepaths, labels, label_id, label_map, transform; items are (img, aug1, aug2, [label,]
basename)).  Images come from a pre-decoded tensor pool: JPEG decoding / CLIP preprocessing is a NEXT row
(SURVEY.md 8f-2)."""
import torch
from torch.utils.data import Dataset


class ImagePool:
    """All decoded images of a dataset directory, shared by its train / val / unlabeled / test views (the
    reference's datasets all open files under one root, so lists can be merged freely across them)."""

    def __init__(self, filepaths, images):
        self.index = {p: i for i, p in enumerate(filepaths)}
        self.images = images


class TensorPoolDataset(Dataset):
    def __init__(self, filepaths, images, root="", train=True, labels=None, label_id=False, label_map=None, transform=None):
        sub = "train" if train else "test"
        self.filepaths = [f"{root}/{sub}/{f}" if root else f for f in filepaths]
        if isinstance(images, ImagePool):
            self._index, self._pool = images.index, images.images
        else:
            self._index = {p: i for i, p in enumerate(self.filepaths)}
            self._pool = images
        self.transform = transform
        self.labels = labels
        self.label_id = label_id
        self.label_map = label_map
        self.train = train

    @property
    def images(self):
        """[N,3,R,R] tensor aligned with the CURRENT filepaths (lists may have been rebuilt by a pseudolabeler)."""
        idx = torch.tensor([self._index[p] for p in self.filepaths], dtype=torch.long)
        return self._pool[idx.to(self._pool.device)]

    @images.setter
    def images(self, value):   # the pseudolabelers reset `images` after rebuilding the lists; the pool itself stays
        pass

    def __len__(self):
        return len(self.filepaths)

    def __getitem__(self, index):
        path = self.filepaths[index]
        img = self._pool[self._index[path]]
        name = path.split("/")[-1]
        if self.labels is not None:
            label = int(self.labels[index]) if self.label_id else int(self.label_map[self.labels[index]])
            return img, img, img, label, name
        return img, img, img, name
