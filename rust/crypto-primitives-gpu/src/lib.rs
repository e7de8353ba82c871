//! `BatchFieldConfig`: the slice-level extension next to the reference's scalar `FieldConfig`
//! (src/field.rs:146-187), served by libfpgpu on an MI355X.  Scalar methods stay on the CPU; every
//! batch method returns exactly what looping the scalar method would (bit-identical canonical
//! Montgomery residues), and fails loudly when no GPU is visible: there is no CPU fallback.
//!
//! NOT COMPILED in the build image (no cargo there): see ../README.md.
//!
//! Elements cross the FFI as contiguous little-endian `u64` words, copied through the accessors
//! the reference already exposes (`Uint::as_words` / `from_words`,
//! src/semiring/crypto_bigint_uint.rs:53-73; `BoxedUint::as_words` / `as_mut_words`,
//! src/semiring/crypto_bigint_boxed_uint.rs:66-78).  No slice of elements is transmuted:
//! crypto-bigint does not promise its structs' layout.

use core::ffi::CStr;
use core::ptr::NonNull;
use std::collections::HashMap;
use std::sync::{Arc, Mutex, OnceLock};

use crypto_primitives::{
    crypto_bigint_boxed_monty::{BoxedMontyField, BoxedMontyFieldElement},
    crypto_bigint_boxed_uint::BoxedUint,
    crypto_bigint_const_monty::ConstMontyField,
    crypto_bigint_monty::{MontyField, MontyFieldElement},
    crypto_bigint_uint::Uint,
    BaseFieldConfig, DenseRowMatrix, FieldConfig, FieldError, FixedConfig, SparseMatrix, Wrapper,
};
use crypto_bigint::modular::ConstMontyParams;
use fpgpu_sys as sys;

/// Why a batch call failed.  `InvalidModulus` mirrors `FieldError::InvalidModulus`.
#[derive(Debug, Clone, PartialEq, Eq)]
pub enum GpuError {
    InvalidModulus,
    Unsupported(String),
    Shape(String),
    Hip(String),
    NoDevice,
}

impl From<GpuError> for FieldError {
    fn from(_: GpuError) -> Self {
        FieldError::InvalidModulus
    }
}

fn last_error() -> String {
    // Safety: never NULL, NUL-terminated, thread-local storage owned by the library.
    unsafe { CStr::from_ptr(sys::fpgpu_last_error_string()) }.to_string_lossy().into_owned()
}

fn check(rc: i32) -> Result<(), GpuError> {
    match rc {
        sys::FPGPU_OK => Ok(()),
        sys::FPGPU_ERR_INVALID_MODULUS => Err(GpuError::InvalidModulus),
        sys::FPGPU_ERR_UNSUPPORTED => Err(GpuError::Unsupported(last_error())),
        sys::FPGPU_ERR_SHAPE => Err(GpuError::Shape(last_error())),
        sys::FPGPU_ERR_NO_DEVICE => Err(GpuError::NoDevice),
        sys::FPGPU_ERR_RCCL => Err(GpuError::Hip(last_error())),
        _ => Err(GpuError::Hip(last_error())),
    }
}

/// Owning handle of an `fpgpu_ctx` (modulus, R, R^2, -p^-1).  Immutable after creation, so it is
/// `Send + Sync` like the reference's configs (src/lib.rs:62-65).
pub struct GpuCtx {
    raw: NonNull<sys::fpgpu_ctx>,
    n64: usize,
}
unsafe impl Send for GpuCtx {}
unsafe impl Sync for GpuCtx {}

impl GpuCtx {
    /// `modulus_le`: the modulus as little-endian 64-bit words (`modulus().as_words()`).
    pub fn new(modulus_le: &[u64]) -> Result<Self, GpuError> {
        let mut raw = core::ptr::null_mut();
        // Safety: `modulus_le` holds `len` words; `raw` receives an owned context or stays null.
        check(unsafe { sys::fpgpu_ctx_create(modulus_le.as_ptr(), modulus_le.len() as u32, &mut raw) })?;
        Ok(Self { raw: NonNull::new(raw).expect("fpgpu_ctx_create returned OK and a null context"), n64: modulus_le.len() })
    }
    pub fn n64(&self) -> usize {
        self.n64
    }
    fn ptr(&self) -> *const sys::fpgpu_ctx {
        self.raw.as_ptr()
    }
}

impl Drop for GpuCtx {
    fn drop(&mut self) {
        // Safety: created by fpgpu_ctx_create, destroyed once.
        unsafe { sys::fpgpu_ctx_destroy(self.raw.as_ptr()) }
    }
}

/// One `GpuCtx` per modulus for the life of the process (what INTEGRATION.md calls the cached context).
fn cached_ctx(modulus_le: &[u64]) -> Result<Arc<GpuCtx>, GpuError> {
    static CACHE: OnceLock<Mutex<HashMap<Vec<u64>, Arc<GpuCtx>>>> = OnceLock::new();
    let cache = CACHE.get_or_init(|| Mutex::new(HashMap::new()));
    let mut map = cache.lock().unwrap_or_else(|poisoned| poisoned.into_inner());
    if let Some(ctx) = map.get(modulus_le) {
        return Ok(Arc::clone(ctx));
    }
    let ctx = Arc::new(GpuCtx::new(modulus_le)?);
    map.insert(modulus_le.to_vec(), Arc::clone(&ctx));
    Ok(ctx)
}

/// How a config's elements turn into words and back.  The three reference adapters implement
/// this; `BatchFieldConfig` is written once on top of it.
pub trait GpuMarshal: FieldConfig {
    /// 64-bit words per element (`LIMBS`, or `bits_precision / 64` for the boxed field).
    fn n64(&self) -> usize;
    /// The modulus as little-endian words.
    fn modulus_words(&self) -> Vec<u64>;
    /// Montgomery representative of `e`, little-endian words, into `out[..n64]`.
    fn write_words(&self, e: &Self::Element, out: &mut [u64]);
    /// The element whose Montgomery representative is `words[..n64]` (a canonical residue).
    fn read_words(&self, words: &[u64]) -> Self::Element;
    /// The exponent type's words (for `pow_batch`).
    fn integer_words(&self, e: &Self::Integer) -> Vec<u64>;

    /// The field's device context, made once per modulus and shared from then on (a context costs
    /// 2 * 64 * n64 modular doublings on the host for R and R^2; it is immutable, `Send + Sync`, and
    /// keyed by the modulus words, which determine it completely).
    fn gpu_ctx(&self) -> Result<Arc<GpuCtx>, GpuError> {
        cached_ctx(&self.modulus_words())
    }
    fn gather(&self, v: &[Self::Element]) -> Vec<u64> {
        let n64 = self.n64();
        let mut w = vec![0u64; v.len() * n64];
        for (e, chunk) in v.iter().zip(w.chunks_exact_mut(n64)) {
            self.write_words(e, chunk);
        }
        w
    }
    fn scatter(&self, w: &[u64]) -> Vec<Self::Element> {
        w.chunks_exact(self.n64()).map(|c| self.read_words(c)).collect()
    }
}

/// Slice-level methods next to `FieldConfig`.  Each is the batched form of the scalar method
/// named in its doc line.
pub trait BatchFieldConfig: GpuMarshal {
    /// `mul` (src/semiring.rs:159): `out[i] = a[i] * b[i]`.
    fn mul_batch(&self, a: &[Self::Element], b: &[Self::Element]) -> Result<Vec<Self::Element>, GpuError> {
        binary(self, a, b, sys::fpgpu_mul)
    }
    /// `add` (src/semiring.rs:153).
    fn add_batch(&self, a: &[Self::Element], b: &[Self::Element]) -> Result<Vec<Self::Element>, GpuError> {
        binary(self, a, b, sys::fpgpu_add)
    }
    /// `sub` (src/semiring.rs:156).
    fn sub_batch(&self, a: &[Self::Element], b: &[Self::Element]) -> Result<Vec<Self::Element>, GpuError> {
        binary(self, a, b, sys::fpgpu_sub)
    }
    /// `neg` (src/ring.rs:45-56).
    fn neg_batch(&self, a: &[Self::Element]) -> Result<Vec<Self::Element>, GpuError> {
        unary(self, a, sys::fpgpu_neg)
    }
    /// `Square` (crypto_bigint_const_monty.rs:736-740).
    fn square_batch(&self, a: &[Self::Element]) -> Result<Vec<Self::Element>, GpuError> {
        unary(self, a, sys::fpgpu_sqr)
    }
    /// `ConstMontyField::double` (crypto_bigint_const_monty.rs:86-89).
    fn double_batch(&self, a: &[Self::Element]) -> Result<Vec<Self::Element>, GpuError> {
        unary(self, a, sys::fpgpu_double)
    }
    /// `ConstMontyField::div_by_2` (crypto_bigint_const_monty.rs:81-84): the `x` with `x + x = a`.
    fn div_by_2_batch(&self, a: &[Self::Element]) -> Result<Vec<Self::Element>, GpuError> {
        unary(self, a, sys::fpgpu_div_by_2)
    }
    /// `inv` (src/field.rs:152): `None` exactly where the scalar method returns `None`.
    fn inv_batch(&self, x: &[Self::Element]) -> Result<Vec<Option<Self::Element>>, GpuError> {
        let ctx = self.gpu_ctx()?;
        let xw = self.gather(x);
        let mut ow = vec![0u64; xw.len()];
        let mut none = vec![0u8; x.len()];
        // Safety: all three buffers hold x.len() elements / flags.
        check(unsafe { sys::fpgpu_batch_inv(ctx.ptr(), xw.as_ptr(), ow.as_mut_ptr(), none.as_mut_ptr(), x.len()) })?;
        Ok(self.scatter(&ow).into_iter().zip(none).map(|(v, f)| (f == 0).then_some(v)).collect())
    }
    /// `div` (src/field.rs:155-157): panics with "Division by zero" if any divisor has no inverse.
    fn div_batch(&self, a: &[Self::Element], b: &[Self::Element]) -> Result<Vec<Self::Element>, GpuError> {
        Ok(self.checked_div_batch(a, b)?.into_iter().map(|o| o.expect("Division by zero")).collect())
    }
    /// `checked_div` (src/field.rs:168-170): `None` exactly where the divisor has no inverse.  One device call.
    fn checked_div_batch(&self, a: &[Self::Element], b: &[Self::Element]) -> Result<Vec<Option<Self::Element>>, GpuError> {
        assert_eq!(a.len(), b.len(), "operand length mismatch");
        let ctx = self.gpu_ctx()?;
        let (aw, bw) = (self.gather(a), self.gather(b));
        let mut ow = vec![0u64; aw.len()];
        let mut none = vec![0u8; a.len()];
        // Safety: all four buffers hold a.len() elements / flags.
        check(unsafe { sys::fpgpu_div(ctx.ptr(), aw.as_ptr(), bw.as_ptr(), ow.as_mut_ptr(), none.as_mut_ptr(), a.len()) })?;
        Ok(self.scatter(&ow).into_iter().zip(none).map(|(v, f)| (f == 0).then_some(v)).collect())
    }
    /// `pow` (src/field.rs:160) with one exponent for the whole batch.
    fn pow_batch(&self, x: &[Self::Element], e: &Self::Integer) -> Result<Vec<Self::Element>, GpuError> {
        let ctx = self.gpu_ctx()?;
        let xw = self.gather(x);
        let ew = self.integer_words(e);
        let mut ow = vec![0u64; xw.len()];
        // Safety: `ew` holds `ew.len()` words (<= 8 is checked by the library).
        check(unsafe { sys::fpgpu_pow(ctx.ptr(), xw.as_ptr(), ew.as_ptr(), ew.len() as u32, ow.as_mut_ptr(), x.len()) })?;
        Ok(self.scatter(&ow))
    }
    /// `sum_refs` (src/semiring.rs:213-226) as one device fold; zero for an empty slice.
    fn sum_batch(&self, v: &[Self::Element]) -> Result<Self::Element, GpuError> {
        fold(self, v, sys::fpgpu_sum)
    }
    /// `product_refs` (src/semiring.rs:232-241); one for an empty slice.
    fn product_batch(&self, v: &[Self::Element]) -> Result<Self::Element, GpuError> {
        fold(self, v, sys::fpgpu_product)
    }
    /// `sum_refs` (src/semiring.rs:213-226) of `mul` over two slices: a dot product.
    fn dot_batch(&self, a: &[Self::Element], b: &[Self::Element]) -> Result<Self::Element, GpuError> {
        if a.len() != b.len() {
            return Err(GpuError::Shape("operand lengths differ".into()));
        }
        let ctx = self.gpu_ctx()?;
        let (aw, bw) = (self.gather(a), self.gather(b));
        let mut ow = vec![0u64; self.n64()];
        // Safety: `aw`, `bw` hold a.len() elements, `ow` one.
        check(unsafe { sys::fpgpu_dot(ctx.ptr(), aw.as_ptr(), bw.as_ptr(), a.len(), ow.as_mut_ptr()) })?;
        Ok(self.read_words(&ow))
    }
    /// `y[i] = sum_refs_j mul(A[i][j], x[j])` over `DenseRowMatrix::as_rows` (src/matrix.rs:162-164).
    fn matvec(&self, a: &DenseRowMatrix<Self::Element>, x: &[Self::Element]) -> Result<Vec<Self::Element>, GpuError> {
        if x.len() != a.num_cols || a.data.len() != a.num_rows * a.num_cols {
            return Err(GpuError::Shape("matrix / vector shape mismatch".into()));
        }
        let ctx = self.gpu_ctx()?;
        let (aw, xw) = (self.gather(&a.data), self.gather(x));
        let mut yw = vec![0u64; a.num_rows * self.n64()];
        // Safety: A holds rows*cols, x cols, y rows elements.
        check(unsafe { sys::fpgpu_matvec(ctx.ptr(), aw.as_ptr(), a.num_rows, a.num_cols, xw.as_ptr(), yw.as_mut_ptr()) })?;
        Ok(self.scatter(&yw))
    }
    /// The same fold over `SparseMatrix` (src/matrix.rs:46-61: `density` cells `(column, value)` per row).
    fn spmv(&self, a: &SparseMatrix<Self::Element>, x: &[Self::Element]) -> Result<Vec<Self::Element>, GpuError> {
        if x.len() != a.num_cols || a.cells.len() != a.num_rows * a.density || a.num_cols > u32::MAX as usize {
            return Err(GpuError::Shape("sparse matrix / vector shape mismatch".into()));
        }
        let ctx = self.gpu_ctx()?;
        let n64 = self.n64();
        let mut vals = vec![0u64; a.cells.len() * n64];
        let mut cols = Vec::with_capacity(a.cells.len());
        for ((c, v), chunk) in a.cells.iter().zip(vals.chunks_exact_mut(n64)) {
            cols.push(u32::try_from(*c).map_err(|_| GpuError::Shape("column index beyond u32".into()))?);
            self.write_words(v, chunk);
        }
        let xw = self.gather(x);
        let mut yw = vec![0u64; a.num_rows * n64];
        // Safety: vals / cols hold rows*density cells, x cols, y rows elements.
        check(unsafe {
            sys::fpgpu_spmv(ctx.ptr(), vals.as_ptr(), cols.as_ptr(), a.num_rows, a.num_cols, a.density, xw.as_ptr(), yw.as_mut_ptr())
        })?;
        Ok(self.scatter(&yw))
    }
    /// `C[i][j] = sum_refs_l mul(A[i][l], B[l][j])`.
    fn matmul(&self, a: &DenseRowMatrix<Self::Element>, b: &DenseRowMatrix<Self::Element>)
              -> Result<DenseRowMatrix<Self::Element>, GpuError> {
        if a.num_cols != b.num_rows || a.data.len() != a.num_rows * a.num_cols || b.data.len() != b.num_rows * b.num_cols {
            return Err(GpuError::Shape("matrix shapes do not chain".into()));
        }
        let ctx = self.gpu_ctx()?;
        let (aw, bw) = (self.gather(&a.data), self.gather(&b.data));
        let mut cw = vec![0u64; a.num_rows * b.num_cols * self.n64()];
        // Safety: A is m x k, B is k x n, C is m x n elements.
        check(unsafe {
            sys::fpgpu_matmul(ctx.ptr(), aw.as_ptr(), bw.as_ptr(), cw.as_mut_ptr(), a.num_rows, a.num_cols, b.num_cols)
        })?;
        Ok(DenseRowMatrix { num_rows: a.num_rows, num_cols: b.num_cols, data: self.scatter(&cw) })
    }
}
impl<T: GpuMarshal> BatchFieldConfig for T {}

// ---- device-resident vectors: the fast path behind the trait ---------------------------------
// A host-slice batch call pays PCIe for every operand of every call (and, for BoxedMontyField, a
// gather over one heap allocation per element, src/field/crypto_bigint_boxed_monty.rs:62-64).  A
// `DeviceVec` keeps the elements in HBM between calls: gather + upload once, chain any number of
// batch operations with no host copy, download + scatter once.  `Placement` lays the vector out
// over the device list of `set_devices`: contiguous ranges for elementwise work and inversion (no
// collective), whole rows per device for the matrix products -- the reference's own parallel axis,
// rayon over `par_chunks_exact(num_cols)` (src/matrix.rs:170-184) -- with an RCCL all-gather when
// the output is replicated.

/// Where a `DeviceVec` lives.
#[derive(Debug, Clone, Copy, PartialEq, Eq)]
pub enum Placement {
    /// The calling thread's current HIP device.
    Current,
    /// Device `i`.
    Device(u32),
    /// Contiguous, balanced ranges (whole rows with `alloc_rows`) over the device list.
    Sharded,
    /// One full copy on every device of the list.
    Replicated,
}
impl Placement {
    fn raw(self) -> i32 {
        match self {
            Placement::Current => sys::FPGPU_PLACE_CURRENT,
            Placement::Device(i) => i as i32,
            Placement::Sharded => sys::FPGPU_PLACE_SHARDED,
            Placement::Replicated => sys::FPGPU_PLACE_REPLICATED,
        }
    }
}

/// The device list that sharded / replicated vectors use (`&[]` returns to "current device alone").
pub fn set_devices(ids: &[i32]) -> Result<(), GpuError> {
    // Safety: `ids` holds `len` device indices.
    check(unsafe { sys::fpgpu_set_devices(ids.as_ptr(), ids.len() as i32) })
}

/// A device list as a value (`fpgpu_devset`): lay vectors out over it with `DeviceVec::alloc_on` /
/// `alloc_rows_on` instead of the process-wide list of `set_devices`.  Vectors keep the list alive.
pub struct DeviceSet {
    raw: NonNull<sys::fpgpu_devset>,
}
unsafe impl Send for DeviceSet {}
unsafe impl Sync for DeviceSet {}
impl DeviceSet {
    pub fn new(ids: &[i32]) -> Result<Self, GpuError> {
        let mut raw = core::ptr::null_mut();
        // Safety: `ids` holds `len` device indices; `raw` receives an owned list or stays null.
        check(unsafe { sys::fpgpu_devset_create(ids.as_ptr(), ids.len() as i32, &mut raw) })?;
        Ok(Self { raw: NonNull::new(raw).expect("fpgpu_devset_create returned OK and a null list") })
    }
}
impl Drop for DeviceSet {
    fn drop(&mut self) {
        // Safety: created by fpgpu_devset_create, destroyed once; vectors laid out over it stay valid.
        unsafe { sys::fpgpu_devset_destroy(self.raw.as_ptr()) }
    }
}

/// Owning handle of an `fpgpu_vec`.  Operations on it are asynchronous on the library's
/// per-device streams; `download` and the value-returning calls synchronise.
pub struct DeviceVec {
    raw: NonNull<sys::fpgpu_vec>,
    n64: usize, // words per element, from the context the vector was made with: the transfers check slice lengths against it
}
unsafe impl Send for DeviceVec {}

impl DeviceVec {
    pub fn alloc(ctx: &GpuCtx, n: usize, place: Placement) -> Result<Self, GpuError> {
        let mut raw = core::ptr::null_mut();
        // Safety: `raw` receives an owned vector or stays null.
        check(unsafe { sys::fpgpu_vec_alloc(ctx.ptr(), n, place.raw(), &mut raw) })?;
        Ok(Self { raw: NonNull::new(raw).expect("fpgpu_vec_alloc returned OK and a null vector"), n64: ctx.n64() })
    }
    /// `rows * cols` elements; a sharded placement gives every device whole rows.
    pub fn alloc_rows(ctx: &GpuCtx, rows: usize, cols: usize, place: Placement) -> Result<Self, GpuError> {
        let mut raw = core::ptr::null_mut();
        check(unsafe { sys::fpgpu_vec_alloc_rows(ctx.ptr(), rows, cols, place.raw(), &mut raw) })?;
        Ok(Self { raw: NonNull::new(raw).expect("fpgpu_vec_alloc_rows returned OK and a null vector"), n64: ctx.n64() })
    }
    /// As `alloc`, laid out over `set` instead of the default device list.
    pub fn alloc_on(ctx: &GpuCtx, set: &DeviceSet, n: usize, place: Placement) -> Result<Self, GpuError> {
        let mut raw = core::ptr::null_mut();
        check(unsafe { sys::fpgpu_vec_alloc_on(ctx.ptr(), set.raw.as_ptr(), n, place.raw(), &mut raw) })?;
        Ok(Self { raw: NonNull::new(raw).expect("fpgpu_vec_alloc_on returned OK and a null vector"), n64: ctx.n64() })
    }
    /// As `alloc_rows`, laid out over `set`.
    pub fn alloc_rows_on(ctx: &GpuCtx, set: &DeviceSet, rows: usize, cols: usize, place: Placement) -> Result<Self, GpuError> {
        let mut raw = core::ptr::null_mut();
        check(unsafe { sys::fpgpu_vec_alloc_rows_on(ctx.ptr(), set.raw.as_ptr(), rows, cols, place.raw(), &mut raw) })?;
        Ok(Self { raw: NonNull::new(raw).expect("fpgpu_vec_alloc_rows_on returned OK and a null vector"), n64: ctx.n64() })
    }
    /// `Ok` iff `words` is exactly `n` elements and `[first, first + n)` lies inside the vector.
    fn check_range(&self, words_len: usize, first: usize, n: usize) -> Result<(), GpuError> {
        let want = n.checked_mul(self.n64).ok_or_else(|| GpuError::Shape("element count overflows".into()))?;
        if words_len != want {
            return Err(GpuError::Shape(format!("slice of {words_len} words for {n} elements of {} words", self.n64)));
        }
        match first.checked_add(n) {
            Some(end) if end <= self.len() => Ok(()),
            _ => Err(GpuError::Shape("range beyond the vector".into())),
        }
    }
    pub fn len(&self) -> usize {
        unsafe { sys::fpgpu_vec_len(self.raw.as_ptr()) }
    }
    pub fn is_empty(&self) -> bool {
        self.len() == 0
    }
    /// `words`: `n * n64` little-endian words (the output of `GpuMarshal::gather`).
    pub fn upload_words(&mut self, words: &[u64], first: usize, n: usize) -> Result<(), GpuError> {
        self.check_range(words.len(), first, n)?;
        // Safety: `words` holds exactly n * n64 words (checked above), the library reads no more.
        check(unsafe { sys::fpgpu_vec_upload(self.raw.as_ptr(), words.as_ptr(), first, n) })
    }
    pub fn download_words(&self, words: &mut [u64], first: usize, n: usize) -> Result<(), GpuError> {
        self.check_range(words.len(), first, n)?;
        // Safety: `words` holds exactly n * n64 words (checked above), the library writes no more.
        check(unsafe { sys::fpgpu_vec_download(self.raw.as_ptr(), words.as_mut_ptr(), first, n) })
    }
    pub fn sync(&self) -> Result<(), GpuError> {
        check(unsafe { sys::fpgpu_vec_sync(self.raw.as_ptr()) })
    }
    /// `self <- src` between any two layouts of the same length (scatter, all-gather, re-shard).
    pub fn copy_from(&mut self, src: &DeviceVec) -> Result<(), GpuError> {
        check(unsafe { sys::fpgpu_vec_copy(self.raw.as_ptr(), src.raw.as_ptr()) })
    }
    fn ptr(&self) -> *const sys::fpgpu_vec {
        self.raw.as_ptr()
    }
    fn ptr_mut(&mut self) -> *mut sys::fpgpu_vec {
        self.raw.as_ptr()
    }
}
impl Drop for DeviceVec {
    fn drop(&mut self) {
        // Safety: created by fpgpu_vec_alloc*, freed once (the library waits for queued work first).
        unsafe { sys::fpgpu_vec_free(self.raw.as_ptr()) }
    }
}

type VBin = unsafe extern "C" fn(*const sys::fpgpu_ctx, *const sys::fpgpu_vec, *const sys::fpgpu_vec, *mut sys::fpgpu_vec) -> i32;
type VUn = unsafe extern "C" fn(*const sys::fpgpu_ctx, *const sys::fpgpu_vec, *mut sys::fpgpu_vec) -> i32;

/// `BatchFieldConfig` on device-resident vectors.  One context per chain (`gpu_ctx()` once), the
/// same reference methods as the slice-level trait, no host copies between calls.
pub trait DeviceBatchFieldConfig: GpuMarshal {
    /// Gather the elements' words once and upload them.
    fn to_device(&self, ctx: &GpuCtx, v: &[Self::Element], place: Placement) -> Result<DeviceVec, GpuError> {
        let mut d = DeviceVec::alloc(ctx, v.len(), place)?;
        d.upload_words(&self.gather(v), 0, v.len())?;
        Ok(d)
    }
    /// `DenseRowMatrix::data` (src/matrix.rs:135-145) with whole rows per device.
    fn matrix_to_device(&self, ctx: &GpuCtx, a: &DenseRowMatrix<Self::Element>, place: Placement) -> Result<DeviceVec, GpuError> {
        let mut d = DeviceVec::alloc_rows(ctx, a.num_rows, a.num_cols, place)?;
        d.upload_words(&self.gather(&a.data), 0, a.data.len())?;
        Ok(d)
    }
    /// Download and scatter back into elements.
    fn from_device(&self, d: &DeviceVec) -> Result<Vec<Self::Element>, GpuError> {
        let mut w = vec![0u64; d.len() * self.n64()];
        d.download_words(&mut w, 0, d.len())?;
        Ok(self.scatter(&w))
    }
    fn mul_dev(&self, ctx: &GpuCtx, a: &DeviceVec, b: &DeviceVec, out: &mut DeviceVec) -> Result<(), GpuError> {
        vbinary(ctx, a, b, out, sys::fpgpu_vec_mul)
    }
    fn add_dev(&self, ctx: &GpuCtx, a: &DeviceVec, b: &DeviceVec, out: &mut DeviceVec) -> Result<(), GpuError> {
        vbinary(ctx, a, b, out, sys::fpgpu_vec_add)
    }
    fn sub_dev(&self, ctx: &GpuCtx, a: &DeviceVec, b: &DeviceVec, out: &mut DeviceVec) -> Result<(), GpuError> {
        vbinary(ctx, a, b, out, sys::fpgpu_vec_sub)
    }
    fn neg_dev(&self, ctx: &GpuCtx, a: &DeviceVec, out: &mut DeviceVec) -> Result<(), GpuError> {
        vunary(ctx, a, out, sys::fpgpu_vec_neg)
    }
    fn square_dev(&self, ctx: &GpuCtx, a: &DeviceVec, out: &mut DeviceVec) -> Result<(), GpuError> {
        vunary(ctx, a, out, sys::fpgpu_vec_sqr)
    }
    fn double_dev(&self, ctx: &GpuCtx, a: &DeviceVec, out: &mut DeviceVec) -> Result<(), GpuError> {
        vunary(ctx, a, out, sys::fpgpu_vec_double)
    }
    fn div_by_2_dev(&self, ctx: &GpuCtx, a: &DeviceVec, out: &mut DeviceVec) -> Result<(), GpuError> {
        vunary(ctx, a, out, sys::fpgpu_vec_div_by_2)
    }
    /// `pow` with one exponent for the whole vector.
    fn pow_dev(&self, ctx: &GpuCtx, a: &DeviceVec, e: &Self::Integer, out: &mut DeviceVec) -> Result<(), GpuError> {
        let ew = self.integer_words(e);
        check(unsafe { sys::fpgpu_vec_pow(ctx.ptr(), a.ptr(), ew.as_ptr(), ew.len() as u32, out.ptr_mut()) })
    }
    /// `inv` on every element; the returned flags are 1 exactly where the scalar method returns
    /// `None` (`out[i]` is zero there).  Every device inverts its own range: no collective.
    fn inv_dev(&self, ctx: &GpuCtx, x: &DeviceVec, out: &mut DeviceVec) -> Result<Vec<bool>, GpuError> {
        let mut none = vec![0u8; x.len()];
        check(unsafe { sys::fpgpu_vec_batch_inv(ctx.ptr(), x.ptr(), out.ptr_mut(), none.as_mut_ptr(), core::ptr::null_mut()) })?;
        Ok(none.into_iter().map(|f| f != 0).collect())
    }
    /// `checked_div` on device vectors: flags as `inv_dev`.
    fn div_dev(&self, ctx: &GpuCtx, a: &DeviceVec, b: &DeviceVec, out: &mut DeviceVec) -> Result<Vec<bool>, GpuError> {
        let mut none = vec![0u8; a.len()];
        check(unsafe { sys::fpgpu_vec_div(ctx.ptr(), a.ptr(), b.ptr(), out.ptr_mut(), none.as_mut_ptr(), core::ptr::null_mut()) })?;
        Ok(none.into_iter().map(|f| f != 0).collect())
    }
    fn sum_dev(&self, ctx: &GpuCtx, v: &DeviceVec) -> Result<Self::Element, GpuError> {
        let mut w = vec![0u64; self.n64()];
        check(unsafe { sys::fpgpu_vec_sum(ctx.ptr(), v.ptr(), w.as_mut_ptr()) })?;
        Ok(self.read_words(&w))
    }
    fn product_dev(&self, ctx: &GpuCtx, v: &DeviceVec) -> Result<Self::Element, GpuError> {
        let mut w = vec![0u64; self.n64()];
        check(unsafe { sys::fpgpu_vec_product(ctx.ptr(), v.ptr(), w.as_mut_ptr()) })?;
        Ok(self.read_words(&w))
    }
    /// `sum_refs` of `mul` over two vectors: one row of the mat-vec.
    fn dot_dev(&self, ctx: &GpuCtx, a: &DeviceVec, b: &DeviceVec) -> Result<Self::Element, GpuError> {
        let mut w = vec![0u64; self.n64()];
        check(unsafe { sys::fpgpu_vec_dot(ctx.ptr(), a.ptr(), b.ptr(), w.as_mut_ptr()) })?;
        Ok(self.read_words(&w))
    }
    /// `y = A x`: A in row blocks (from `matrix_to_device(.., Placement::Sharded)`), `x` whole on
    /// every device (`Replicated`), `y` sharded like A's rows (no communication) or `Replicated`
    /// (all-gather of the row blocks over xGMI).
    fn matvec_dev(&self, ctx: &GpuCtx, a: &DeviceVec, rows: usize, cols: usize, x: &DeviceVec, y: &mut DeviceVec)
                  -> Result<(), GpuError> {
        check(unsafe { sys::fpgpu_vec_matvec(ctx.ptr(), a.ptr(), rows, cols, x.ptr(), y.ptr_mut()) })
    }
    fn matmul_dev(&self, ctx: &GpuCtx, a: &DeviceVec, b: &DeviceVec, c: &mut DeviceVec, m: usize, k: usize, n: usize)
                  -> Result<(), GpuError> {
        check(unsafe { sys::fpgpu_vec_matmul(ctx.ptr(), a.ptr(), b.ptr(), c.ptr_mut(), m, k, n) })
    }
}
impl<T: GpuMarshal> DeviceBatchFieldConfig for T {}

fn vbinary(ctx: &GpuCtx, a: &DeviceVec, b: &DeviceVec, out: &mut DeviceVec, f: VBin) -> Result<(), GpuError> {
    // Safety: three live vectors of one context's element width; layouts are checked by the library.
    check(unsafe { f(ctx.ptr(), a.ptr(), b.ptr(), out.ptr_mut()) })
}
fn vunary(ctx: &GpuCtx, a: &DeviceVec, out: &mut DeviceVec, f: VUn) -> Result<(), GpuError> {
    check(unsafe { f(ctx.ptr(), a.ptr(), out.ptr_mut()) })
}

type Bin = unsafe extern "C" fn(*const sys::fpgpu_ctx, *const u64, *const u64, *mut u64, usize) -> i32;
type Un = unsafe extern "C" fn(*const sys::fpgpu_ctx, *const u64, *mut u64, usize) -> i32;
type Fold = unsafe extern "C" fn(*const sys::fpgpu_ctx, *const u64, usize, *mut u64) -> i32;

fn binary<C: GpuMarshal + ?Sized>(c: &C, a: &[C::Element], b: &[C::Element], f: Bin) -> Result<Vec<C::Element>, GpuError> {
    if a.len() != b.len() {
        return Err(GpuError::Shape("operand lengths differ".into()));
    }
    let ctx = c.gpu_ctx()?;
    let (aw, bw) = (c.gather(a), c.gather(b));
    let mut ow = vec![0u64; aw.len()];
    // Safety: the three buffers hold a.len() elements of n64 words.
    check(unsafe { f(ctx.ptr(), aw.as_ptr(), bw.as_ptr(), ow.as_mut_ptr(), a.len()) })?;
    Ok(c.scatter(&ow))
}

fn unary<C: GpuMarshal + ?Sized>(c: &C, a: &[C::Element], f: Un) -> Result<Vec<C::Element>, GpuError> {
    let ctx = c.gpu_ctx()?;
    let aw = c.gather(a);
    let mut ow = vec![0u64; aw.len()];
    // Safety: both buffers hold a.len() elements of n64 words.
    check(unsafe { f(ctx.ptr(), aw.as_ptr(), ow.as_mut_ptr(), a.len()) })?;
    Ok(c.scatter(&ow))
}

fn fold<C: GpuMarshal + ?Sized>(c: &C, v: &[C::Element], f: Fold) -> Result<C::Element, GpuError> {
    let ctx = c.gpu_ctx()?;
    let vw = c.gather(v);
    let mut ow = vec![0u64; c.n64()];
    // Safety: `vw` holds v.len() elements, `ow` one.
    check(unsafe { f(ctx.ptr(), vw.as_ptr(), v.len(), ow.as_mut_ptr()) })?;
    Ok(c.read_words(&ow))
}

// ---- MontyField<LIMBS> (src/field/crypto_bigint_monty.rs:16-58): runtime modulus, stack limbs
impl<const LIMBS: usize> GpuMarshal for MontyField<LIMBS> {
    fn n64(&self) -> usize {
        LIMBS
    }
    fn modulus_words(&self) -> Vec<u64> {
        self.modulus().as_words().to_vec() // BaseFieldConfig::modulus, :355-357
    }
    fn write_words(&self, e: &MontyFieldElement<LIMBS>, out: &mut [u64]) {
        out.copy_from_slice(e.0.as_words()); // the Montgomery representative (:56-58)
    }
    fn read_words(&self, words: &[u64]) -> MontyFieldElement<LIMBS> {
        let mut w = [0u64; LIMBS];
        w.copy_from_slice(words);
        MontyFieldElement(Uint::from_words(w))
    }
    fn integer_words(&self, e: &Uint<LIMBS>) -> Vec<u64> {
        e.as_words().to_vec()
    }
}

// ---- BoxedMontyField (src/field/crypto_bigint_boxed_monty.rs:18-64): heap limbs, one
// allocation per element, so the adapter gathers them into one contiguous buffer.
impl GpuMarshal for BoxedMontyField {
    fn n64(&self) -> usize {
        (self.bits_precision() / 64) as usize // R = 2^bits_precision (:55-57)
    }
    fn modulus_words(&self) -> Vec<u64> {
        self.modulus().as_words().to_vec()
    }
    fn write_words(&self, e: &BoxedMontyFieldElement, out: &mut [u64]) {
        out.copy_from_slice(e.0.as_words());
    }
    fn read_words(&self, words: &[u64]) -> BoxedMontyFieldElement {
        let mut v = BoxedUint::new(crypto_bigint::BoxedUint::zero_with_precision(self.bits_precision()));
        v.as_mut_words().copy_from_slice(words);
        BoxedMontyFieldElement(v)
    }
    fn integer_words(&self, e: &BoxedUint) -> Vec<u64> {
        e.as_words().to_vec()
    }
}

// ---- FixedConfig<ConstMontyField<Mod, LIMBS>> (src/lib.rs:78-89,
// src/field/crypto_bigint_const_monty.rs:27-31): compile-time modulus
impl<Mod: ConstMontyParams<LIMBS>, const LIMBS: usize> GpuMarshal for FixedConfig<ConstMontyField<Mod, LIMBS>>
where
    FixedConfig<ConstMontyField<Mod, LIMBS>>: FieldConfig<Element = ConstMontyField<Mod, LIMBS>, Integer = Uint<LIMBS>>,
{
    fn n64(&self) -> usize {
        LIMBS
    }
    fn modulus_words(&self) -> Vec<u64> {
        Mod::PARAMS.modulus().as_ref().as_words().to_vec()
    }
    fn write_words(&self, e: &ConstMontyField<Mod, LIMBS>, out: &mut [u64]) {
        out.copy_from_slice(e.as_montgomery().as_words()); // :61-64
    }
    fn read_words(&self, words: &[u64]) -> ConstMontyField<Mod, LIMBS> {
        let mut w = [0u64; LIMBS];
        w.copy_from_slice(words);
        ConstMontyField::from_montgomery(Uint::from_words(w)) // :71-74
    }
    fn integer_words(&self, e: &Uint<LIMBS>) -> Vec<u64> {
        e.as_words().to_vec()
    }
}

/// `BaseFieldConfig::new` with the modulus check done by the library: an even modulus is
/// rejected by both sides with `InvalidModulus` (crypto_bigint_monty.rs:346-352).
pub fn monty_field_with_gpu<const LIMBS: usize>(modulus: &Uint<LIMBS>) -> Result<(MontyField<LIMBS>, Arc<GpuCtx>), FieldError> {
    let ctx = cached_ctx(modulus.as_words())?;
    Ok((MontyField::<LIMBS>::new(modulus)?, ctx))
}

#[cfg(test)]
mod tests {
    //! The reference's own test shapes (known answers on secp256k1's prime,
    //! crypto_bigint_monty.rs:564-592; inv(0) = None, :778-780), through the batch methods,
    //! against the scalar methods of the same config.  Needs an MI355X.
    use super::*;
    use crypto_primitives::{ProjectElementWithConfig, RingConfig, SemiringConfig};

    fn field() -> MontyField<4> {
        let p = Uint::<4>::from_words([0xFFFFFFFEFFFFFC2F, u64::MAX, u64::MAX, u64::MAX]); // 2^256 - 2^32 - 977
        MontyField::<4>::new(&p).unwrap()
    }

    #[test]
    fn batch_equals_scalar() {
        let f = field();
        let a: Vec<_> = (1u64..200).map(|v| f.project(&v.wrapping_mul(0x9E3779B97F4A7C15))).collect();
        let b: Vec<_> = (1u64..200).map(|v| f.project(&v.wrapping_mul(0xBF58476D1CE4E5B9))).collect();
        let mul = f.mul_batch(&a, &b).unwrap();
        let add = f.add_batch(&a, &b).unwrap();
        let sub = f.sub_batch(&a, &b).unwrap();
        let neg = f.neg_batch(&a).unwrap();
        for i in 0..a.len() {
            assert_eq!(mul[i], f.mul(&a[i], &b[i]));
            assert_eq!(add[i], f.add(&a[i], &b[i]));
            assert_eq!(sub[i], f.sub(&a[i], &b[i]));
            assert_eq!(neg[i], f.neg(&a[i]));
        }
        assert_eq!(f.sum_batch(&a).unwrap(), f.sum_refs(a.iter()));
        assert_eq!(f.product_batch(&a).unwrap(), f.product_refs(a.iter()));
    }

    #[test]
    fn inversion_and_zero() {
        let f = field();
        let mut x: Vec<_> = (1u64..100).map(|v| f.project(&v)).collect();
        x[7] = f.zero();
        let inv = f.inv_batch(&x).unwrap();
        for (i, v) in x.iter().enumerate() {
            assert_eq!(inv[i], f.inv(v));
        }
        assert!(inv[7].is_none());
    }

    #[test]
    fn device_chain_equals_scalar() {
        // a*b + a, squared, inverted: three dependent batch operations on HBM-resident vectors,
        // one upload per operand and one download; sharded over a list that names GPU 0 twice
        let f = field();
        let ctx = f.gpu_ctx().unwrap();
        let a: Vec<_> = (1u64..500).map(|v| f.project(&v.wrapping_mul(0x9E3779B97F4A7C15))).collect();
        let b: Vec<_> = (1u64..500).map(|v| f.project(&v.wrapping_mul(0xBF58476D1CE4E5B9))).collect();
        for ids in [&[][..], &[0][..], &[0, 0][..]] {
            set_devices(ids).unwrap();
            let place = if ids.is_empty() { Placement::Current } else { Placement::Sharded };
            let (da, db) = (f.to_device(&ctx, &a, place).unwrap(), f.to_device(&ctx, &b, place).unwrap());
            let mut t = DeviceVec::alloc(&ctx, a.len(), place).unwrap();
            let mut u = DeviceVec::alloc(&ctx, a.len(), place).unwrap();
            f.mul_dev(&ctx, &da, &db, &mut t).unwrap();
            f.add_dev(&ctx, &t, &da, &mut u).unwrap();
            f.square_dev(&ctx, &u, &mut t).unwrap();
            let none = f.inv_dev(&ctx, &t, &mut u).unwrap();
            let got = f.from_device(&u).unwrap();
            for i in 0..a.len() {
                let s = f.add(&f.mul(&a[i], &b[i]), &a[i]);
                let want = f.inv(&f.mul(&s, &s));
                assert_eq!(none[i], want.is_none());
                assert_eq!(got[i], want.unwrap_or_else(|| f.zero()));
            }
            assert_eq!(f.dot_dev(&ctx, &da, &db).unwrap(), f.dot_batch(&a, &b).unwrap());
        }
        set_devices(&[]).unwrap();
    }

    #[test]
    #[should_panic(expected = "Division by zero")]
    fn division_by_zero_panics() {
        let f = field();
        let _ = f.div_batch(&[f.one()], &[f.zero()]);
    }

    #[test]
    fn short_slices_are_rejected_not_read() {
        let f = field();
        let ctx = f.gpu_ctx().unwrap();
        let mut d = DeviceVec::alloc(&ctx, 8, Placement::Current).unwrap();
        assert!(matches!(d.upload_words(&[0u64; 8 * 4 - 1], 0, 8), Err(GpuError::Shape(_))));
        assert!(matches!(d.upload_words(&[0u64; 4], 8, 1), Err(GpuError::Shape(_))));
        let mut short = [0u64; 3];
        assert!(matches!(d.download_words(&mut short, 0, 1), Err(GpuError::Shape(_))));
        assert!(Arc::ptr_eq(&ctx, &f.gpu_ctx().unwrap())); // one context per modulus
    }

    #[test]
    fn even_modulus_rejected() {
        assert_eq!(GpuCtx::new(&[42, 0, 0, 0]).err(), Some(GpuError::InvalidModulus));
    }
}
