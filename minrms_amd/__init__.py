"""MI355X-native engine for the MINRMS exhaustive RMSD-vs-N search."""
