"""Names only: the reference builds transform pipelines it never applies to
tensor-backed datasets."""


class _T:
    def __init__(self, *a, **k):
        self.a, self.k = a, k

    def __call__(self, x):
        return x


class Compose(_T):
    def __call__(self, x):
        for t in self.a[0]:
            x = t(x)
        return x


Scale = Resize = CenterCrop = ToTensor = Normalize = RandomHorizontalFlip = _T
RandomSizedCrop = RandomResizedCrop = RandomCrop = Lambda = ToPILImage = _T
