"""The reference's own tests of the hot-path helpers (tests/test_utils.py:58-63 and :72-159 of JoHof/lungmask v0.2.20), restated
against `lungmask_amd.utils` -- same inputs, same expected values."""
import numpy as np

from lungmask_amd import utils


def check_reference_utils_tests(engine):
    utils.set_engine(engine)
    try:
        # test_bbox_3D (test_utils.py:58-63)
        m = np.zeros((10, 10, 10), dtype=np.uint8)
        m[2:8, 3:7, 4:6] = 1
        bb = utils.bbox_3D(m, margin=2)
        assert tuple(bb) == (0, 10, 1, 9, 2, 8)
        # test_simple_bodymask (test_utils.py:72-78)
        img = np.full((10, 10), dtype=np.int16, fill_value=-1000)
        img[2:8, 3:7] = 1
        img[9, 9] = 1
        assert np.sum(utils.simple_bodymask(img)) == 24
        # test_crop_and_resize (:81-88)
        cropped, bb = utils.crop_and_resize(img, width=20, height=20)
        assert tuple(bb) == (2, 3, 8, 7) and cropped.shape == (20, 20) and np.sum(cropped) == 400 and cropped.dtype == np.int16
        # test_preprocess (:91-99)
        vol = np.full((2, 10, 10), dtype=np.int16, fill_value=-1000)
        vol[:, 2:8, 3:7] = 1
        vol[:, 9, 9] = 1
        cropped, boxes = utils.preprocess(vol, resolution=[20, 20])
        for sl, bb_ in zip(cropped, boxes):
            assert tuple(bb_) == (2, 3, 8, 7) and sl.shape == (20, 20) and np.sum(sl) == 400
        # test_reshape_mask (:102-107)
        msk = np.full((10, 10), dtype=np.uint8, fill_value=1)
        cropped_mask = utils.reshape_mask(msk, (2, 2, 22, 22), origsize=(30, 30))
        assert cropped_mask.shape == (30, 30) and np.sum(cropped_mask) == 400
        # test_postprocessing (:124-159)
        label_image = np.zeros((1, 6, 6), dtype=np.uint8)
        label_image[0] = np.asarray([[0, 0, 0, 0, 0, 0], [0, 1, 1, 2, 2, 0], [0, 2, 0, 3, 1, 0], [0, 4, 4, 4, 0, 0], [0, 4, 0, 4, 0, 0], [0, 4, 4, 4, 0, 0]])
        res_gt = [[0, 0, 0, 0, 0, 0], [0, 1, 1, 2, 2, 0], [0, 1, 0, 3, 2, 0], [0, 4, 4, 4, 0, 0], [0, 4, 0, 4, 0, 0], [0, 4, 4, 4, 0, 0]]
        res = utils.postprocessing(np.tile(label_image, (2, 1, 1)), spare=[], disable_tqdm=True, skip_below=1)[0]
        assert np.all(res == res_gt)
        res = utils.postprocessing(np.tile(label_image, (2, 1, 1)), spare=[3], disable_tqdm=True, skip_below=1)[0]
        assert res[2, 3] == 2
        res = utils.postprocessing(np.tile(label_image, (2, 1, 1)), spare=[3], disable_tqdm=True, skip_below=3)[0]
        assert res[2, 1] == 0
        # argument checking of the mirror
        try:
            utils.crop_and_resize(np.full((10, 10), 3000, np.int16))
            raise AssertionError("unclipped input must be refused")
        except ValueError:
            pass
        try:
            utils.preprocess(np.zeros((1, 10, 10), np.float32))
            raise AssertionError("float input must be refused")
        except TypeError:
            pass
    finally:
        utils.set_engine(None)
