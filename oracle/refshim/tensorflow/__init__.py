"""tensorflow stand-in: big_vision/utils.py does `import tensorflow.io.gfile as gfile` at module level."""
