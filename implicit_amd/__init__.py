"""implicit_amd: MI355X-native ALS training + top-k scoring behind benfred/implicit's `implicit.gpu`
plug-in surface.  See DESIGN.md and INTEGRATION.md."""
__version__ = "0.1.0"
