def equivalent_layers(*a, **k):  # shadowed by the reference's own definition
    raise NotImplementedError
