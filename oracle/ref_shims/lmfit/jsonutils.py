def encode4js(obj):
    return obj


def decode4js(obj):
    return obj
