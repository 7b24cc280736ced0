def get_layer_dims(layers):
    """[a, b, c] -> [(a, b), (b, c)]   (reference utils/util.py:273-275)"""
    return list(zip(layers[:-1], layers[1:]))
