"""absl.app stand-in (imported by the reference's trainer module at module level)."""
def run(main): raise RuntimeError("the trainer's main() is not run here")
