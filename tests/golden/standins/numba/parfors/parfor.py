def ensure_parallel_support():
    pass
