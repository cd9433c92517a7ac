"""Drop-in mirror of the reference's ``lib`` package for the channel-pruning hot path
(lib/decompose.py, lib/net.py, lib/worker.py, lib/cfgs.py of ethanhe42/channel-pruning),
with the arithmetic running on MI355X through libcpmi355.so."""
