def DEFINE_config_file(*a, **k): pass
