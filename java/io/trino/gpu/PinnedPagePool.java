package io.trino.gpu;

import io.trino.spi.Page;
import io.trino.spi.block.Block;
import io.trino.spi.block.ByteArrayBlock;
import io.trino.spi.block.DictionaryBlock;
import io.trino.spi.block.IntArrayBlock;
import io.trino.spi.block.LongArrayBlock;
import io.trino.spi.block.RunLengthEncodedBlock;
import io.trino.spi.block.VariableWidthBlock;
import io.trino.spi.type.Type;

import java.nio.ByteBuffer;
import java.nio.ByteOrder;
import java.util.ArrayList;
import java.util.List;

/**
 * Stages Pages into pinned direct ByteBuffers (pa_host_malloc_pinned) for GpuNative.addInput: JVM heap arrays are movable, so
 * every Block's arrays are copied out once (SURVEY 8b "Ownership").  Buffers are 4 MiB slabs filled back to back: consecutive
 * small pages of one operator continue each other in pinned memory, and, with the slabs kept until the operator is closed,
 * are handed over as PA_PAGE_STABLE -- the native side then gathers them without per-page work.
 */
final class PinnedPagePool
{
    static final class StagedPage
    {
        int positions;
        int channels;
        int[] types;
        int[] encodings;          // pa_encoding: 0 FLAT, 1 VARWIDTH, 2 DICTIONARY, 3 RLE
        long[] valueOffsets;
        long[] offsetOffsets;
        long[] nullOffsets;
        long[] idOffsets;
        int[] dictionaryChannel;
        int[] dictionarySize;
        ByteBuffer buffer;
        boolean stable;
    }

    private static final int SLAB_BYTES = 4 << 20;
    private final List<Type> types;
    private final List<ByteBuffer> slabs;
    private ByteBuffer current;

    PinnedPagePool(List<Type> types)
    {
        this.types = types;
        this.slabs = new ArrayList<>();
    }

    StagedPage stage(Page page)
    {
        int channels = page.getChannelCount();
        List<Block> blocks = new ArrayList<>();
        for (int c = 0; c < channels; c++) {
            blocks.add(page.getBlock(c));
        }
        // dictionaries travel as extra "channels" behind the page's own
        int[] dictionaryChannel = new int[channels * 2];
        java.util.Arrays.fill(dictionaryChannel, -1);
        for (int c = 0; c < channels; c++) {
            Block b = blocks.get(c);
            if (b instanceof DictionaryBlock) {
                dictionaryChannel[c] = blocks.size();
                blocks.add(((DictionaryBlock) b).getDictionary());
            }
            else if (b instanceof RunLengthEncodedBlock) {
                dictionaryChannel[c] = blocks.size();
                blocks.add(((RunLengthEncodedBlock) b).getValue());
            }
        }
        long bytes = 0;
        for (Block b : blocks) {
            bytes += b.getSizeInBytes() + 64L * 4 + 4L * (b.getPositionCount() + 1);
        }
        ByteBuffer target = reserve(bytes);
        StagedPage s = new StagedPage();
        int total = blocks.size();
        s.positions = page.getPositionCount();
        s.channels = channels;
        s.types = new int[total];
        s.encodings = new int[total];
        s.valueOffsets = new long[total];
        s.offsetOffsets = new long[total];
        s.nullOffsets = new long[total];
        s.idOffsets = new long[total];
        s.dictionaryChannel = java.util.Arrays.copyOf(dictionaryChannel, total);
        s.dictionarySize = new int[total];
        s.buffer = target;
        s.stable = true; // the slab is kept until releaseAll()
        for (int i = 0; i < total; i++) {
            Block b = blocks.get(i);
            Type type = i < channels ? types.get(i) : types.get(indexOf(dictionaryChannel, i));
            s.types[i] = RowExpressionSerializer.typeOf(type);
            s.valueOffsets[i] = s.offsetOffsets[i] = s.nullOffsets[i] = s.idOffsets[i] = -1;
            int n = b.getPositionCount();
            if (b instanceof DictionaryBlock) {
                s.encodings[i] = 2;
                s.idOffsets[i] = align(target);
                for (int p = 0; p < n; p++) {
                    target.putInt(((DictionaryBlock) b).getId(p));
                }
                s.dictionarySize[i] = ((DictionaryBlock) b).getDictionary().getPositionCount();
                continue;
            }
            if (b instanceof RunLengthEncodedBlock) {
                s.encodings[i] = 3;
                s.dictionarySize[i] = 1;
                continue;
            }
            if (b.mayHaveNull()) {
                s.nullOffsets[i] = align(target);
                for (int p = 0; p < n; p++) {
                    target.put((byte) (b.isNull(p) ? 1 : 0));   // boolean[] valueIsNull, 1 B / position
                }
            }
            if (b instanceof LongArrayBlock) {
                s.valueOffsets[i] = align(target);
                for (int p = 0; p < n; p++) {
                    target.putLong(b.getLong(p, 0));
                }
            }
            else if (b instanceof IntArrayBlock) {
                s.valueOffsets[i] = align(target);
                for (int p = 0; p < n; p++) {
                    target.putInt(b.getInt(p, 0));
                }
            }
            else if (b instanceof ByteArrayBlock) {
                s.valueOffsets[i] = align(target);
                for (int p = 0; p < n; p++) {
                    target.put(b.getByte(p, 0));
                }
            }
            else if (b instanceof VariableWidthBlock) {
                s.encodings[i] = 1;
                s.offsetOffsets[i] = align(target);
                int at = 0;
                target.putInt(0);
                for (int p = 0; p < n; p++) {
                    at += b.getSliceLength(p);
                    target.putInt(at);
                }
                s.valueOffsets[i] = align(target);
                for (int p = 0; p < n; p++) {
                    target.put(b.getSlice(p, 0, b.getSliceLength(p)).getBytes());
                }
            }
            else {
                throw new RowExpressionSerializer.UnsupportedOnDevice("block " + b.getClass().getSimpleName());
            }
        }
        return s;
    }

    private static int indexOf(int[] values, int value)
    {
        for (int i = 0; i < values.length; i++) {
            if (values[i] == value) {
                return i;
            }
        }
        throw new IllegalArgumentException();
    }

    /** 16-byte aligned position inside the slab (the native side reads 16 bytes per lane). */
    private static long align(ByteBuffer target)
    {
        target.position((target.position() + 15) & ~15);
        return target.position();
    }

    private ByteBuffer reserve(long bytes)
    {
        if (current == null || current.remaining() < bytes + 16) {
            current = GpuNative.hostMallocPinned(Math.max(SLAB_BYTES, bytes + 16)).order(ByteOrder.LITTLE_ENDIAN);
            slabs.add(current);
        }
        return current;
    }

    /**
     * A page of a page source: the blocks that are loaded are staged as stage() does; an unloaded LazyBlock gets its type and encoding
     * only (the native side asks for it through GpuPageSource.loadBlock when an expression needs it).  The slab of a scan's page is
     * reused for the next page: the native operator has consumed a page when it asks for the next one.
     */
    StagedPage stageLoadedBlocks(Page page)
    {
        Block[] blocks = new Block[page.getChannelCount()];
        for (int c = 0; c < blocks.length; c++) {
            Block b = page.getBlock(c);
            boolean loaded = !(b instanceof io.trino.spi.block.LazyBlock) || b.isLoaded();
            // an unloaded block is staged as an empty block of its channel's kind: positions are the page's, arrays absent
            blocks[c] = loaded ? b.getLoadedBlock() : types.get(c).createBlockBuilder(null, 0).build();
        }
        if (current != null) {
            current.clear();          // (the previous page of this source was consumed)
        }
        StagedPage s = stage(new Page(page.getPositionCount(), blocks));
        s.stable = false;
        return s;
    }

    /** One block staged on its own (a LazyBlock the native side asked for): the addresses of [values, offsets, nulls], 0 = absent. */
    long[] stageBlock(int channel, Block loaded)
    {
        List<Type> saved = new ArrayList<>(types);
        StagedPage s = new PinnedPagePool(java.util.Collections.singletonList(saved.get(channel)), this).stage(new Page(loaded.getPositionCount(), loaded));
        long base = GpuNative.bufferAddress(s.buffer);
        return new long[] {s.valueOffsets[0] >= 0 ? base + s.valueOffsets[0] : 0, s.offsetOffsets[0] >= 0 ? base + s.offsetOffsets[0] : 0,
                s.nullOffsets[0] >= 0 ? base + s.nullOffsets[0] : 0};
    }

    /** A view of `parent` for one channel: stages into the parent's slabs (they are released with the parent). */
    private PinnedPagePool(List<Type> types, PinnedPagePool parent)
    {
        this.types = types;
        this.slabs = parent.slabs;
        this.current = parent.current;
    }

    void close()
    {
        releaseAll();
    }

    void releaseAll()
    {
        for (ByteBuffer slab : slabs) {
            GpuNative.hostFreePinned(slab);
        }
        slabs.clear();
        current = null;
    }
}
