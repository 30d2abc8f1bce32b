"""imported by big_vision/utils.py at module level, unused on this path"""
