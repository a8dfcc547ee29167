"""Drop-in counterparts of the reference's `nets` package for the gaitset hot path."""
