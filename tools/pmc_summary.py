"""Per-kernel summary of a rocprofv3 --pmc counter_collection.csv (one counter per row)."""
import csv, sys, collections, re
sys.path.insert(0, __file__.rsplit("/", 1)[0])
from trace_gaps import short  # noqa
