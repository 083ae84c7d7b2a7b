"""Import-compatibility package: the reference's `Configs` import path over this build's `branchgan` package (a user of the reference switches nothing but sys.path)."""
