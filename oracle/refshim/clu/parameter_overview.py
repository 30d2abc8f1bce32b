def log_parameter_overview(*a, **k): pass
