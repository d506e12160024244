"""Import-path shim: the reference's package name `core`, backed by griduniverse_amd (see compat/README.md)."""
