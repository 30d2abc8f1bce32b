"""jax.lax: nothing on the executed path uses it (models/common.py names it inside a decode-only branch)."""
