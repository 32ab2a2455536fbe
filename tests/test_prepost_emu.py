"""CPU suite: pre-processing / un-crop kernel SOURCES executed through the tests/emu
emulator, bit-exact against goldens produced by the reference's utils.py."""
import prepost_cases as cases


def test_preprocess_emulated_bit_exact(emu_engine):
    assert cases.check_preprocess(emu_engine) >= 8


def test_reshape_mask_emulated_bit_exact(emu_engine):
    assert cases.check_reshape(emu_engine) >= 10
