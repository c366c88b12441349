"""deepspeech.pytorch_amd -- MI355X-native DeepSpeech2 train-step hot path (drop-in for
``deepspeech_pytorch.model.DeepSpeech``).  See DESIGN.md / INTEGRATION.md."""
