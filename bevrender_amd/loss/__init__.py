"""Drop-in counterparts of the reference's `loss` package for the correlation half of the hot path."""
