registry = {}


def register(id, **kw):
    registry[id] = kw
