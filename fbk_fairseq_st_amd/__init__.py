"""MI355X-native (gfx950) speech-translation hot path behind the fairseq plug-in surface of
FBK-fairseq-ST's `conv_transformer`.  Importing the package does not touch the GPU."""
__version__ = "0.1.0"
