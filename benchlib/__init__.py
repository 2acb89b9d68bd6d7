"""The parts of bench.py that are not its timed loop: constants, roofline blocks, CPU baseline / oracle parity, the compact stdout line."""
