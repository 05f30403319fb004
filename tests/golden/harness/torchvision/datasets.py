import torch.utils.data as data


class ImageFolder(data.Dataset):
    def __init__(self, *a, **k):
        raise NotImplementedError("harness stub: no image folders in the container")


class folder:
    IMG_EXTENSIONS = (".jpg", ".jpeg", ".png")

    @staticmethod
    def default_loader(path):
        raise NotImplementedError
