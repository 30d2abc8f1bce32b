"""clu stand-in: the reference's trainer imports clu.parameter_overview at module level (logging only)."""
