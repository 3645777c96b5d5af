"""Drop-in module names of the reference (`models.*`)."""
