"""Dataset plumbing is out of scope; placeholders so the reference package imports."""


class Data:
    def __init__(self, **kw):
        for k, v in kw.items():
            setattr(self, k, v)


class InMemoryDataset:
    def __init__(self, *a, **k):
        raise NotImplementedError("dataset loaders are out of scope (no network)")


def download_url(*a, **k):
    raise NotImplementedError("no network")
