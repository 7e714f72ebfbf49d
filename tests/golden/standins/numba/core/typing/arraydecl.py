def get_array_index_type(*_a, **_k):
    raise NotImplementedError
