"""BASELINE configs[3] at the bench's own shape: bench.ReceiverBank (the object `bench.py --workload
receivers` times) stepped three times over its 2^22-sample ADC block with (a) a slice of the
receiver set that straddles a rank boundary (receivers 123..132 of the 1024: every zoom 1..10, ranks 0
and 1 of the 8 x 128 sharding) and (b) rank 1's whole slice of 128 receivers, and EVERY stage of EVERY receiver checked against the oracle fed the same stream
with its state carried from step to step: both DDCs bit-exact on all their output, the frame the
waterfall took, the u8 row, the wf_pkt_t, the unpacked audio samples, CFastFIR, CAgc mono16 and the
ADPCM payload."""
import os
import sys

import pytest

pytestmark = pytest.mark.gpu

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

STEPS = 3


@pytest.mark.parametrize("NR,FIRST", [(10, 123),      # every zoom once, straddling the rank 0 / rank 1 boundary
                                      (128, 128)])    # rank 1's whole slice: the per-GPU shape of configs[3]
def test_receiver_bank_every_stage_every_receiver(gpu_ctx, oracle, NR, FIRST):
    """bench.check_receiver_bank is the stage-by-stage comparison (also run by `bench.py --workload receivers` on a
    sample of its receivers after the timed region); here on EVERY receiver of the bank."""
    import torch
    import bench
    dev = torch.device("cuda", 0)
    bank = bench.ReceiverBank(0, dev, NR, 1 << 22, FIRST)
    try:
        assert sorted(set(p.zoom for p in bank.params)) == list(range(1, 11))
        got = bench.check_receiver_bank(bank, range(NR), STEPS)
        assert got == {"receivers": NR, "steps": STEPS, "audio_blocks": 2 * NR}     # 402 records a step: a 512-sample block on steps 2 and 3
        assert bank.counts == {"frames": STEPS * NR, "audio_blocks": 2 * NR}
    finally:
        bank.close()
