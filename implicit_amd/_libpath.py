"""Where `implicit_amd.gpu` finds libimplicit_hip.so.  The product always loads the in-tree build; measurement tooling (bench.py with
IMP_LIB_PATH, profiles/scripts/*.sh) may point OVERRIDE at a compile-time variant of the same library BEFORE importing
implicit_amd.gpu.  No environment variable is read here."""
OVERRIDE = None
