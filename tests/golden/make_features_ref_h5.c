#include <hdf5.h>
#include <stdint.h>
#include <stdlib.h>
int main(int argc, char** argv) {
    hid_t f = H5Fcreate(argv[1], H5F_ACC_TRUNC, H5P_DEFAULT, H5P_DEFAULT);
    hsize_t d1[1] = {5};
    int64_t ids[5] = {139, 285, 632, 724, 776};
    hid_t s = H5Screate_simple(1, d1, NULL);
    hid_t ds = H5Dcreate2(f, "image_ids", H5T_NATIVE_INT64, s, H5P_DEFAULT, H5P_DEFAULT, H5P_DEFAULT);
    H5Dwrite(ds, H5T_NATIVE_INT64, H5S_ALL, H5S_ALL, H5P_DEFAULT, ids);
    H5Dclose(ds); H5Sclose(s);
    hsize_t d3[3] = {5, 6, 16};
    float* g = malloc(5 * 6 * 16 * 4);
    for (int i = 0; i < 5 * 6 * 16; ++i) g[i] = (float)(i % 97) * 0.25f - 3.0f;
    s = H5Screate_simple(3, d3, NULL);
    ds = H5Dcreate2(f, "gri_feat", H5T_NATIVE_FLOAT, s, H5P_DEFAULT, H5P_DEFAULT, H5P_DEFAULT);
    H5Dwrite(ds, H5T_NATIVE_FLOAT, H5S_ALL, H5S_ALL, H5P_DEFAULT, g);
    H5Dclose(ds); H5Sclose(s);
    /* numpy bool as h5py stores it: enum over int8, FALSE = 0, TRUE = 1 */
    hid_t bt = H5Tenum_create(H5T_NATIVE_INT8);
    int8_t v = 0; H5Tenum_insert(bt, "FALSE", &v); v = 1; H5Tenum_insert(bt, "TRUE", &v);
    hsize_t d4[4] = {5, 1, 1, 6};
    int8_t m[30];
    for (int i = 0; i < 30; ++i) m[i] = (i % 7 == 3) || (i % 5 == 0);
    s = H5Screate_simple(4, d4, NULL);
    ds = H5Dcreate2(f, "gri_mask", bt, s, H5P_DEFAULT, H5P_DEFAULT, H5P_DEFAULT);
    H5Dwrite(ds, bt, H5S_ALL, H5S_ALL, H5P_DEFAULT, m);
    H5Dclose(ds); H5Sclose(s);
    hsize_t d3b[3] = {5, 4, 8};
    float* r = malloc(5 * 4 * 8 * 4);
    for (int i = 0; i < 160; ++i) r[i] = (float)(i * i % 31) - 15.5f;
    s = H5Screate_simple(3, d3b, NULL);
    ds = H5Dcreate2(f, "reg_feat", H5T_NATIVE_FLOAT, s, H5P_DEFAULT, H5P_DEFAULT, H5P_DEFAULT);
    H5Dwrite(ds, H5T_NATIVE_FLOAT, H5S_ALL, H5S_ALL, H5P_DEFAULT, r);
    H5Dclose(ds); H5Sclose(s);
    hsize_t d4b[4] = {5, 1, 1, 4};
    int8_t rm[20] = {0};
    s = H5Screate_simple(4, d4b, NULL);
    ds = H5Dcreate2(f, "reg_mask", bt, s, H5P_DEFAULT, H5P_DEFAULT, H5P_DEFAULT);
    H5Dwrite(ds, bt, H5S_ALL, H5S_ALL, H5P_DEFAULT, rm);
    H5Dclose(ds); H5Sclose(s); H5Tclose(bt);
    H5Fclose(f);
    return 0;
}
