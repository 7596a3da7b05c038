"""Drop-in counterparts of the reference's `model` package (same module and class names)."""
