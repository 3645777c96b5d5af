"""Drop-in module names of the reference (`utils.*`): thin re-exports of vitta_amd."""
