"""navierstokes_amd — MI355X-native CSR SpMV / matrix-powers path behind the
reference's mpk/SpMV.h interface.  Importing the package does
not load the HIP library; `navierstokes_amd.mpk` does, and fails loudly if it
is missing."""
__version__ = "0.1.0"
