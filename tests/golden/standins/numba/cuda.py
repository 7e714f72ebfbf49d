def is_available():
    return False
