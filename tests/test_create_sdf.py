"""Host logic of the batch driver (reference test/create_sdf.py surface): output layout, view sources."""
import os

import numpy as np
import pytest

from disn_amd import create_sdf


def test_output_layout_matches_reference():
    # reference test/create_sdf.py:87-93: test_objs/<RESOLUTION>_<iso>, "camest_" prefix with --cam_est
    assert create_sdf.result_obj_path("ck/exp", 64, 0.0, False) == os.path.join("ck/exp", "test_objs", "65_0.0")
    assert create_sdf.result_obj_path("ck/exp", 256, 0.003, True) == os.path.join("ck/exp", "test_objs", "camest_257_0.003")
    # :305-311: <dir>/<cat>/<cat>_<obj>_<view:02d>.obj
    assert create_sdf.obj_file_name("r", "03001627", "abc", 7) == os.path.join("r", "03001627", "03001627_abc_07.obj")
    assert create_sdf.obj_file_name("r", "03001627", "abc", "11") == os.path.join("r", "03001627", "03001627_abc_11.obj")


def test_categories_are_the_reference_table():
    assert len(create_sdf.CATS) == 13 and create_sdf.CATS["chair"] == "03001627" and create_sdf.CATS["watercraft"] == "04530566"
    assert create_sdf.ALL_CATS[0] == "airplane" and create_sdf.ALL_CATS[-1] == "watercraft"


def test_prepare_image_follows_h5_convention():
    rng = np.random.default_rng(0)
    raw = rng.integers(0, 256, (137, 137, 4), dtype=np.uint8)
    img = create_sdf.prepare_image(raw)
    assert img.shape == (137, 137, 3) and img.dtype == np.float32
    assert np.array_equal(img, raw[:, :, :3].astype(np.float32) / np.float32(255.0))
    with pytest.raises(ValueError):
        create_sdf.prepare_image(raw[:, :, 0])


def test_synthetic_views_are_seeded_and_cover_non_unit_boxes():
    a = list(create_sdf.synthetic_views(6))
    b = list(create_sdf.synthetic_views(6))
    assert [v["obj_nm"] for v in a] == ["synthetic%06d" % i for i in range(6)]
    for va, vb in zip(a, b):
        assert np.array_equal(va["img"], vb["img"]) and np.array_equal(va["sdf_params"], vb["sdf_params"])
    assert a[3]["cat_id"] == create_sdf.CATS[create_sdf.ALL_CATS[3]]
    assert np.array_equal(a[0]["sdf_params"], [-1, -1, -1, 1, 1, 1])
    p = a[2]["sdf_params"]
    assert not np.array_equal(p, [-1, -1, -1, 1, 1, 1]) and bool((p[3:] > p[:3]).all())


def test_npz_views_round_trip(tmp_path):
    rng = np.random.default_rng(1)
    raw = rng.integers(0, 256, (137, 137, 4), dtype=np.uint8)
    tm = rng.standard_normal((4, 3)).astype(np.float32)
    np.savez(tmp_path / "v0.npz", img_arr=raw, trans_mat=tm, sdf_params=np.array([-.5, -.6, -.7, .5, .6, .7]),
             cat_id="02691156", obj_nm="plane1", view_id=5)
    (v,) = list(create_sdf.npz_views(str(tmp_path)))
    assert v["cat_id"] == "02691156" and v["obj_nm"] == "plane1" and v["view_id"] == 5
    assert np.array_equal(v["trans_mat"], tm) and v["img"].shape == (137, 137, 3)
    with pytest.raises(SystemExit):
        list(create_sdf.npz_views(str(tmp_path / "empty")))


def test_list_files(tmp_path):
    (tmp_path / "03001627_test.lst").write_text("objA\nobjB\n")
    info = create_sdf.read_list(str(tmp_path), ["03001627"], 2, seed=0)
    assert len(info) == 4 and {i[1] for i in info} == {"objA", "objB"} and all(0 <= i[2] < 24 for i in info)


def test_h5_source_fails_loudly_without_h5py():
    try:
        import h5py  # noqa: F401
    except ImportError:
        with pytest.raises(SystemExit):
            list(create_sdf.h5_views("a", "b", [("c", "o", 0)], False))


def test_demo_reads_the_reference_image_like_cv2():
    """tests/golden/demo_chair_20.png is the reference's own demo input (demo/03001627_..._20.png, data, RGBA).
    cv2.imread(IMREAD_UNCHANGED)[:, :, :3] / 255 (reference demo/demo.py:263-264) = the raw colour channels in
    BGR order, alpha dropped, no compositing."""
    from PIL import Image
    from disn_amd import demo
    path = os.path.join(os.path.dirname(__file__), "golden", "demo_chair_20.png")
    img = demo.read_image(path)
    raw = np.asarray(Image.open(path))
    assert raw.shape == (137, 137, 4)
    assert img.shape == (137, 137, 3) and img.dtype == np.float32
    assert np.array_equal(img[:, :, 0], raw[:, :, 2].astype(np.float32) / np.float32(255.0))   # B
    assert np.array_equal(img[:, :, 2], raw[:, :, 0].astype(np.float32) / np.float32(255.0))   # R
    assert 0.0 <= float(img.min()) and float(img.max()) <= 1.0 and float(img.std()) > 0.01
