"""gfile on the local file system (utils.npload reads checkpoints through it)."""
import os
def GFile(path, mode="r"): return open(path, mode)
def exists(p): return os.path.exists(p)
def makedirs(p): os.makedirs(p, exist_ok=True)
