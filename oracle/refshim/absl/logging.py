"""absl.logging stand-in: the reference's model files log at import / load time only."""
def info(*a, **k): pass
def warning(*a, **k): pass
def error(*a, **k): pass
def debug(*a, **k): pass
def log_first_n(*a, **k): pass
INFO = 0
