"""Drop-in module names of the reference (`corpus.*`)."""
